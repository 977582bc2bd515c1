"""N4: bag / image-directory reader for the offline replay, no ROS installation needed.

The reference is driven by `rosbag play` (launch/*.launch:8; the bags themselves are not in the repository): camera frames
arrive as sensor_msgs/CompressedImage on /pylon_camera/image_raw/compressed and the scale comes from marta_msgs/Altitude on
/drivers/altitude (src/visual_Odometry.cpp:190-196).  This module reads the rosbag 2.0 container directly (records,
chunks with `none`, `bz2` or `lz4` compression, connection, index-data and chunk-info records: sequentially, or through the
index section at `index_pos` so that chunks outside a time range / without a wanted topic are never decompressed),
deserialises ANY message type from the message definition
text each connection carries (so marta_msgs needs no generated code), decodes JPEG / PNG payloads with Pillow on the host,
and returns what visual_odometry_amd.replay.replay() takes.  `BagWriter` exists for tests and for making small fixtures.

Format reference: the rosbag "Bags/Format/2.0" specification (record = header_len, header fields `len name=value`,
data_len, data; op codes 2 message, 3 bag header, 4 index, 5 chunk, 6 chunk info, 7 connection) and the ROS 1 wire
serialisation (little endian, strings and variable arrays with a uint32 count).
"""
from __future__ import annotations

import bz2
import io
import os
import struct

import numpy as np

MAGIC = b"#ROSBAG V2.0\n"
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 2, 3, 4, 5, 6, 7

_PRIM = {"bool": "?", "int8": "b", "uint8": "B", "byte": "b", "char": "B", "int16": "h", "uint16": "H", "int32": "i",
         "uint32": "I", "int64": "q", "uint64": "Q", "float32": "f", "float64": "d"}

HEADER_DEF = "uint32 seq\ntime stamp\nstring frame_id\n"


class Stamp(float):
    """A ROS time as seconds (a float, ~0.2 us resolution at epoch scale) that also keeps the integer (sec, nsec) it was read
    from: differences of stamps - the node's dt, visual_Odometry.cpp:470 - should be taken with `ns`, which is exact."""

    def __new__(cls, sec: int, nsec: int):
        obj = super().__new__(cls, sec + nsec * 1e-9)
        obj.sec, obj.nsec = int(sec), int(nsec)
        return obj

    @property
    def ns(self) -> int:
        return self.sec * 1_000_000_000 + self.nsec


# ------------------------------------------------------------------------------------------------ LZ4 (chunk compression "lz4")
def _xxh32(data: bytes, seed: int = 0) -> int:
    """XXH32 (the checksum of the LZ4 frame format), plain Python: used for the 1-byte header checksum always and for the
    content / block checksums when the xxhash module is absent."""
    P1, P2, P3, P4, P5, M = 2654435761, 2246822519, 3266489917, 668265263, 374761393, 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
    n, i = len(data), 0
    if n >= 16:
        v = [(seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M]
        while i + 16 <= n:
            for k, w in enumerate(struct.unpack_from("<4I", data, i)):
                v[k] = (rotl((v[k] + w * P2) & M, 13) * P1) & M
            i += 16
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
    else:
        h = (seed + P5) & M
    h = (h + n) & M
    while i + 4 <= n:
        h = (rotl((h + struct.unpack_from("<I", data, i)[0] * P3) & M, 17) * P4) & M
        i += 4
    while i < n:
        h = (rotl((h + data[i] * P5) & M, 11) * P1) & M
        i += 1
    h ^= h >> 15
    h = (h * P2) & M
    h ^= h >> 13
    h = (h * P3) & M
    return h ^ (h >> 16)


def _xxh32_fast(data: bytes) -> int:
    try:
        import xxhash

        return xxhash.xxh32(data, seed=0).intdigest()
    except ImportError:
        return _xxh32(data)


def lz4_block_decode(src, max_out: int | None = None) -> bytes:
    """One LZ4 block; truncated or inconsistent input raises ValueError (never IndexError), and the declared maximum is
    checked after the final literals too."""
    try:
        out = _lz4_block_decode(src, max_out)
    except IndexError:
        raise ValueError("lz4: truncated block") from None
    if max_out is not None and len(out) > max_out:
        raise ValueError("lz4: block larger than declared")
    return out


def _lz4_block_decode(src, max_out: int | None = None) -> bytes:
    """One LZ4 block (the sequence format of the LZ4 block specification): token = literal length << 4 | match length - 4,
    lengths of 15 continue in 255-valued bytes, little-endian 16-bit match offset; matches may overlap their own output."""
    src = bytes(src)
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        tok = src[i]
        i += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = src[i]
                i += 1
                ll += b
                if b != 255:
                    break
        if i + ll > n:
            raise ValueError("lz4: truncated block")
        out += src[i : i + ll]
        i += ll
        if i >= n:
            break  # the last sequence has literals only
        off = src[i] | (src[i + 1] << 8)
        i += 2
        if off == 0 or off > len(out):
            raise ValueError("lz4: match offset outside the decoded data")
        ml = (tok & 15) + 4
        if (tok & 15) == 15:
            while True:
                b = src[i]
                i += 1
                ml += b
                if b != 255:
                    break
        start = len(out) - off
        if off >= ml:
            out += out[start : start + ml]
        else:  # overlapping match: the pattern of `off` bytes repeats
            pat = bytes(out[start:])
            out += (pat * (ml // off + 1))[:ml]
        if max_out is not None and len(out) > max_out:
            raise ValueError("lz4: block larger than declared")
    return bytes(out)


def lz4_frame_decode(src) -> bytes:
    """LZ4 frame format (magic 0x184D2204; what roslz4 writes into `lz4` chunks and the lz4 command line tool reads): FLG / BD /
    optional content size and dictionary id / header checksum, then blocks (size word, high bit = stored uncompressed, optional
    block checksum) up to the end mark, optional xxhash32 content checksum.  Header, block and content checksums are all
    verified (xxhash module when present, else the plain-Python XXH32 above); truncated input raises ValueError."""
    try:
        return _lz4_frame_decode(src)
    except (IndexError, struct.error):
        raise ValueError("lz4: truncated frame") from None


def _lz4_frame_decode(src) -> bytes:
    mv = memoryview(bytes(src))
    if len(mv) < 7 or struct.unpack_from("<I", mv, 0)[0] != 0x184D2204:
        raise ValueError("lz4: not an LZ4 frame")
    flg, bd = mv[4], mv[5]
    if (flg >> 6) != 1:
        raise ValueError("lz4: unsupported frame version")
    block_checksum, content_size, content_checksum, dict_id = (flg >> 4) & 1, (flg >> 3) & 1, (flg >> 2) & 1, flg & 1
    block_max = {4: 1 << 16, 5: 1 << 18, 6: 1 << 20, 7: 1 << 22}.get((bd >> 4) & 7)
    if block_max is None:
        raise ValueError("lz4: bad block size code")
    off = 6
    declared = None
    if content_size:
        declared = struct.unpack_from("<Q", mv, off)[0]
        off += 8
    if dict_id:
        off += 4
    if ((_xxh32(bytes(mv[4:off])) >> 8) & 0xFF) != mv[off]:  # HC: second byte of XXH32 over the frame descriptor
        raise ValueError("lz4: header checksum mismatch")
    off += 1
    out = bytearray()
    while True:
        size = struct.unpack_from("<I", mv, off)[0]
        off += 4
        if size == 0:
            break
        stored = bool(size >> 31)
        size &= 0x7FFFFFFF
        data = mv[off : off + size]
        if len(data) != size:
            raise ValueError("lz4: truncated block")
        if block_checksum and _xxh32_fast(bytes(data)) != struct.unpack_from("<I", mv, off + size)[0]:
            raise ValueError("lz4: block checksum mismatch")
        off += size + (4 if block_checksum else 0)
        out += bytes(data) if stored else lz4_block_decode(data, block_max)
    if content_checksum:
        want = struct.unpack_from("<I", mv, off)[0]
        if _xxh32_fast(bytes(out)) != want:
            raise ValueError("lz4: content checksum mismatch")
    if declared is not None and declared != len(out):
        raise ValueError("lz4: content size mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------------ message definitions
class MsgCodec:
    """(De)serialiser built from a connection's message_definition text."""

    def __init__(self, msg_type: str, definition: str):
        self.msg_type = msg_type
        self.types: dict[str, list[tuple[str, str, int | None]]] = {}
        parts = definition.split("\n" + "=" * 80 + "\n")
        self.types[msg_type] = self._parse(parts[0])
        for p in parts[1:]:
            head, _, body = p.partition("\n")
            if head.startswith("MSG:"):
                self.types[head[4:].strip()] = self._parse(body)
        self.types.setdefault("std_msgs/Header", self._parse(HEADER_DEF))

    @staticmethod
    def _parse(text):
        fields = []
        for line in text.splitlines():
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            toks = line.split()
            if len(toks) < 2 or "=" in toks[1] or (len(toks) > 2 and toks[2].startswith("=")):
                continue  # constant
            ftype, name = toks[0], toks[1]
            count = None  # scalar
            if ftype.endswith("]"):
                base, _, n = ftype[:-1].partition("[")
                ftype, count = base, (int(n) if n else -1)  # -1 = variable length
            fields.append((ftype, name, count))
        return fields

    def _resolve(self, ftype, scope):
        if ftype in self.types:
            return ftype
        if ftype == "Header":
            return "std_msgs/Header"
        pkg = scope.split("/")[0]
        if f"{pkg}/{ftype}" in self.types:
            return f"{pkg}/{ftype}"
        for k in self.types:
            if k.endswith("/" + ftype):
                return k
        raise KeyError(f"message type {ftype} is not described by the connection's definition")

    # ---- decode
    def decode(self, buf: bytes):
        val, off = self._dec_type(self.msg_type, memoryview(buf), 0)
        return val

    def _dec_one(self, ftype, scope, mv, off):
        if ftype in _PRIM:
            fmt = "<" + _PRIM[ftype]
            return struct.unpack_from(fmt, mv, off)[0], off + struct.calcsize(fmt)
        if ftype == "string":
            n = struct.unpack_from("<I", mv, off)[0]
            return bytes(mv[off + 4:off + 4 + n]).decode("utf-8", "replace"), off + 4 + n
        if ftype == "time":
            s, ns = struct.unpack_from("<II", mv, off)
            return Stamp(s, ns), off + 8
        if ftype == "duration":
            s, ns = struct.unpack_from("<ii", mv, off)
            return s + ns * 1e-9, off + 8
        return self._dec_type(self._resolve(ftype, scope), mv, off)

    def _dec_type(self, tname, mv, off):
        out = {}
        for ftype, name, count in self.types[tname]:
            if count is None:
                out[name], off = self._dec_one(ftype, tname, mv, off)
                continue
            n = count
            if n < 0:
                n = struct.unpack_from("<I", mv, off)[0]
                off += 4
            if ftype in ("uint8", "char"):
                out[name] = bytes(mv[off:off + n])
                off += n
            elif ftype in _PRIM:
                dt = np.dtype("<" + {"?": "u1"}.get(_PRIM[ftype], np.dtype(_PRIM[ftype]).str[1:]))
                out[name] = np.frombuffer(mv, dt, n, off).copy()
                off += n * dt.itemsize
            else:
                items = []
                for _ in range(n):
                    v, off = self._dec_one(ftype, tname, mv, off)
                    items.append(v)
                out[name] = items
        return out, off

    # ---- encode (tests / fixtures)
    def encode(self, msg: dict) -> bytes:
        b = io.BytesIO()
        self._enc_type(self.msg_type, msg, b)
        return b.getvalue()

    def _enc_one(self, ftype, scope, v, b):
        if ftype in _PRIM:
            b.write(struct.pack("<" + _PRIM[ftype], v))
        elif ftype == "string":
            raw = v.encode("utf-8")
            b.write(struct.pack("<I", len(raw)) + raw)
        elif ftype in ("time", "duration"):
            if isinstance(v, Stamp):
                s, ns = v.sec, v.nsec
            else:
                s = int(np.floor(v))
                ns = int(round((v - s) * 1e9))
            b.write(struct.pack("<II" if ftype == "time" else "<ii", s, ns))
        else:
            self._enc_type(self._resolve(ftype, scope), v, b)

    def _enc_type(self, tname, msg, b):
        for ftype, name, count in self.types[tname]:
            v = msg[name]
            if count is None:
                self._enc_one(ftype, tname, v, b)
                continue
            if count < 0:
                b.write(struct.pack("<I", len(v)))
            if ftype in ("uint8", "char"):
                b.write(bytes(v))
            elif ftype in _PRIM:
                b.write(np.asarray(v, np.dtype("<" + np.dtype(_PRIM[ftype]).str[1:])).tobytes())
            else:
                for item in v:
                    self._enc_one(ftype, tname, item, b)


# ------------------------------------------------------------------------------------------------ container
def _read_header(mv, off):
    hlen = struct.unpack_from("<I", mv, off)[0]
    off += 4
    end, fields = off + hlen, {}
    while off < end:
        flen = struct.unpack_from("<I", mv, off)[0]
        raw = bytes(mv[off + 4:off + 4 + flen])
        name, _, value = raw.partition(b"=")
        fields[name.decode()] = value
        off += 4 + flen
    return fields, end


def _records(mv, off, end):
    while off + 8 <= end:
        fields, off = _read_header(mv, off)
        dlen = struct.unpack_from("<I", mv, off)[0]
        yield fields, mv[off + 4:off + 4 + dlen]
        off += 4 + dlen


class Connection:
    def __init__(self, conn_id, topic, hdr):
        self.id, self.topic = conn_id, topic
        self.type = hdr.get("type", b"").decode()
        self.md5sum = hdr.get("md5sum", b"").decode()
        self.definition = hdr.get("message_definition", b"").decode("utf-8", "replace")
        self._codec = None

    @property
    def codec(self) -> MsgCodec:
        if self._codec is None:
            self._codec = MsgCodec(self.type, self.definition)
        return self._codec


class BagReader:
    """Sequential reader: for topic, conn, t, raw in BagReader(path).messages(topics): ...  (t in seconds, bag receive time)."""

    def __init__(self, path):
        self.path = path
        self.data = np.memmap(path, np.uint8, "r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        mv = memoryview(self.data)
        if bytes(mv[:len(MAGIC)]) != MAGIC:
            raise ValueError(f"{path}: not a rosbag 2.0 file")
        self.connections: dict[int, Connection] = {}
        fields, data = next(_records(mv, len(MAGIC), len(mv)))
        if fields.get("op") != bytes([OP_BAG_HEADER]):
            raise ValueError("first record is not the bag header")
        self.index_pos = struct.unpack("<Q", fields["index_pos"])[0]
        self.conn_count = struct.unpack("<I", fields["conn_count"])[0]
        self.chunk_count = struct.unpack("<I", fields["chunk_count"])[0]

    def _add_connection(self, fields, data):
        cid = struct.unpack("<I", fields["conn"])[0]
        if cid not in self.connections:
            hdr, _ = _read_header(memoryview(struct.pack("<I", len(data)) + bytes(data)), 0)
            self.connections[cid] = Connection(cid, fields["topic"].decode(), hdr)

    @staticmethod
    def _chunk_payload(fields, data):
        comp = fields["compression"].decode()
        if comp == "none":
            inner = data
        elif comp == "bz2":
            inner = memoryview(bz2.decompress(bytes(data)))
        elif comp == "lz4":
            inner = memoryview(lz4_frame_decode(data))
        else:
            raise NotImplementedError(f"chunk compression '{comp}' (none, bz2 and lz4 are what rosbag writes)")
        if "size" in fields and struct.unpack("<I", fields["size"])[0] != len(inner):
            raise ValueError("chunk: uncompressed size does not match the record's `size` field")
        return inner

    def _scan(self, mv, off, end, topics, t_lo=None, t_hi=None):
        for fields, data in _records(mv, off, end):
            op = fields["op"][0]
            if op == OP_CONNECTION:
                self._add_connection(fields, data)
            elif op == OP_MSG:
                conn = self.connections[struct.unpack("<I", fields["conn"])[0]]
                if topics is None or conn.topic in topics:
                    st = Stamp(*struct.unpack("<II", fields["time"]))
                    if (t_lo is None or st.ns >= t_lo) and (t_hi is None or st.ns <= t_hi):
                        yield conn.topic, conn, st, bytes(data)
            elif op == OP_CHUNK:
                inner = self._chunk_payload(fields, data)
                yield from self._scan(inner, 0, len(inner), topics, t_lo, t_hi)
            # OP_INDEX / OP_CHUNK_INFO records carry no messages: the sequential scan passes over them (any other op code,
            # e.g. from a later format revision, is skipped the same way: a record is self-delimiting)

    # ---- the index section (everything from index_pos on: one connection record per connection, one chunk-info per chunk)
    def read_index(self):
        """Connection records and chunk infos from the index section: [(chunk_pos, start_ns, end_ns, {conn: count}), ...] in
        file order.  Raises if the bag was not closed (index_pos = 0: `rosbag reindex` territory; messages() still works)."""
        if self.index_pos == 0:
            raise ValueError("bag has no index section (index_pos = 0: not closed / needs reindexing)")
        mv = memoryview(self.data)
        infos = []
        for fields, data in _records(mv, self.index_pos, len(mv)):
            op = fields["op"][0]
            if op == OP_CONNECTION:
                self._add_connection(fields, data)
            elif op == OP_CHUNK_INFO:
                if struct.unpack("<I", fields["ver"])[0] != 1:
                    raise NotImplementedError("chunk info version != 1")
                n = struct.unpack("<I", fields["count"])[0]
                counts = dict(struct.unpack_from("<II", data, 8 * i) for i in range(n))
                t0 = Stamp(*struct.unpack("<II", fields["start_time"])).ns
                t1 = Stamp(*struct.unpack("<II", fields["end_time"])).ns
                infos.append((struct.unpack("<Q", fields["chunk_pos"])[0], t0, t1, counts))
        return infos

    def chunk_index(self, chunk_pos: int):
        """The index-data records (op 4, version 1) that follow the chunk at chunk_pos: {conn: [(time_ns, offset in the
        uncompressed chunk), ...]}."""
        mv = memoryview(self.data)
        recs = _records(mv, chunk_pos, len(mv))
        fields, _ = next(recs)
        if fields["op"][0] != OP_CHUNK:
            raise ValueError("no chunk record at this position")
        out = {}
        for fields, data in recs:
            if fields["op"][0] != OP_INDEX:
                break
            if struct.unpack("<I", fields["ver"])[0] != 1:
                raise NotImplementedError("index data version != 1")
            n = struct.unpack("<I", fields["count"])[0]
            ent = []
            for i in range(n):
                s_, ns_, off_ = struct.unpack_from("<III", data, 12 * i)
                ent.append((s_ * 1_000_000_000 + ns_, off_))
            out[struct.unpack("<I", fields["conn"])[0]] = ent
        return out

    def messages_indexed(self, topics=None, start=None, end=None):
        """As messages(), but through the index section: only chunks that hold a wanted connection and overlap [start, end]
        (seconds, bag receive time) are read and decompressed."""
        infos = self.read_index()
        want = None if topics is None else {c.id for c in self.connections.values() if c.topic in set(topics)}
        t_lo = None if start is None else int(round(start * 1e9))
        t_hi = None if end is None else int(round(end * 1e9))
        mv = memoryview(self.data)
        for pos, t0, t1, counts in infos:
            if want is not None and not (want & set(counts)):
                continue
            if (t_lo is not None and t1 < t_lo) or (t_hi is not None and t0 > t_hi):
                continue
            fields, data = next(_records(mv, pos, len(mv)))
            inner = self._chunk_payload(fields, data)
            yield from self._scan(inner, 0, len(inner), None if topics is None else set(topics), t_lo, t_hi)

    def messages(self, topics=None):
        mv = memoryview(self.data)
        first = next(_records(mv, len(MAGIC), len(mv)))  # skip the (padded) bag header record
        hlen = struct.unpack_from("<I", mv, len(MAGIC))[0]
        off = len(MAGIC) + 4 + hlen + 4 + len(first[1])
        yield from self._scan(mv, off, len(mv), set(topics) if topics is not None else None)

    def info(self):
        counts: dict[str, int] = {}
        t0 = t1 = None
        for topic, conn, t, _ in self.messages():
            counts[topic] = counts.get(topic, 0) + 1
            t0 = t if t0 is None else min(t0, t)
            t1 = t if t1 is None else max(t1, t)
        return {"topics": {c.topic: (c.type, counts.get(c.topic, 0)) for c in self.connections.values()}, "start": t0, "end": t1}


class BagWriter:
    """Minimal rosbag 2.0 writer (tests, fixtures): one chunk per `chunk_bytes` of messages, index + chunk-info records."""

    def __init__(self, path, compression="none", chunk_bytes=1 << 20):
        assert compression in ("none", "bz2")
        self.f = open(path, "wb")
        self.compression, self.chunk_bytes = compression, chunk_bytes
        self.conns: dict[str, tuple[int, bytes, bytes]] = {}
        self.chunk = io.BytesIO()
        self.chunk_index: dict[int, list[tuple[int, int, int]]] = {}
        self.chunk_infos = []
        self.f.write(MAGIC)
        self._write_bag_header(0, 0, 0)

    @staticmethod
    def _rec(fields: dict, data: bytes) -> bytes:
        h = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v for k, v in fields.items())
        return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data

    def _write_bag_header(self, index_pos, conn_count, chunk_count):
        fields = {"op": bytes([OP_BAG_HEADER]), "index_pos": struct.pack("<Q", index_pos), "conn_count": struct.pack("<I", conn_count),
                  "chunk_count": struct.pack("<I", chunk_count)}
        h = self._rec(fields, b"")
        pad = 4096 - len(h)
        self.f.seek(len(MAGIC))
        self.f.write(h[:-4] + struct.pack("<I", pad) + b" " * pad)

    def add_connection(self, topic, msg_type, definition, md5sum="*"):
        if topic not in self.conns:
            cid = len(self.conns)
            chdr = {"topic": topic.encode(), "type": msg_type.encode(), "md5sum": md5sum.encode(), "message_definition": definition.encode()}
            data = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v for k, v in chdr.items())
            rec = self._rec({"op": bytes([OP_CONNECTION]), "conn": struct.pack("<I", cid), "topic": topic.encode()}, data)
            self.conns[topic] = (cid, rec, b"")
            self.chunk.write(rec)
        return self.conns[topic][0]

    def write(self, topic, t, raw: bytes):
        cid = self.conns[topic][0]
        s = int(np.floor(t))
        tt = struct.pack("<II", s, int(round((t - s) * 1e9)))
        self.chunk_index.setdefault(cid, []).append((struct.unpack("<Q", tt)[0], self.chunk.tell(), t))
        self.chunk.write(self._rec({"op": bytes([OP_MSG]), "conn": struct.pack("<I", cid), "time": tt}, raw))
        if self.chunk.tell() >= self.chunk_bytes:
            self._flush()

    def _flush(self):
        raw = self.chunk.getvalue()
        if not self.chunk_index:
            return
        pos = self.f.tell()
        body = bz2.compress(raw) if self.compression == "bz2" else raw
        self.f.write(self._rec({"op": bytes([OP_CHUNK]), "compression": self.compression.encode(), "size": struct.pack("<I", len(raw))}, body))
        times = []
        for cid, entries in self.chunk_index.items():
            data = b"".join(struct.pack("<QI", tt, off) for tt, off, _ in entries)
            self.f.write(self._rec({"op": bytes([OP_INDEX]), "ver": struct.pack("<I", 1), "conn": struct.pack("<I", cid),
                                    "count": struct.pack("<I", len(entries))}, data))
            times += [(tt & 0xFFFFFFFF) * 1_000_000_000 + (tt >> 32) for tt, _, _ in entries]  # (sec, nsec) words -> nanoseconds
        self.chunk_infos.append((pos, min(times), max(times), {c: len(e) for c, e in self.chunk_index.items()}))
        self.chunk, self.chunk_index = io.BytesIO(), {}

    def close(self):
        self._flush()
        index_pos = self.f.tell()
        for cid, rec, _ in self.conns.values():
            self.f.write(rec)
        for pos, t0, t1, counts in self.chunk_infos:
            data = b"".join(struct.pack("<II", c, n) for c, n in counts.items())
            self.f.write(self._rec({"op": bytes([OP_CHUNK_INFO]), "ver": struct.pack("<I", 1), "chunk_pos": struct.pack("<Q", pos),
                                    "start_time": struct.pack("<II", t0 // 1_000_000_000, t0 % 1_000_000_000),
                                    "end_time": struct.pack("<II", t1 // 1_000_000_000, t1 % 1_000_000_000),
                                    "count": struct.pack("<I", len(counts))}, data))
        self._write_bag_header(index_pos, len(self.conns), len(self.chunk_infos))
        self.f.close()


# ------------------------------------------------------------------------------------------------ images
COMPRESSED_IMAGE_DEF = "Header header\nstring format\nuint8[] data\n" + "=" * 80 + "\nMSG: std_msgs/Header\n" + HEADER_DEF
IMAGE_DEF = ("Header header\nuint32 height\nuint32 width\nstring encoding\nuint8 is_bigendian\nuint32 step\nuint8[] data\n"
             + "=" * 80 + "\nMSG: std_msgs/Header\n" + HEADER_DEF)


def decode_image(msg: dict, msg_type: str) -> np.ndarray:
    """The frame as cv_bridge's toCvCopy(msg, "bgr8") hands it to the node: H x W x 3 uint8, BGR (src/visual_Odometry.cpp:120)."""
    if msg_type.endswith("CompressedImage"):
        from PIL import Image

        im = Image.open(io.BytesIO(msg["data"]))
        rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[..., ::-1])
    h, w, enc, step = msg["height"], msg["width"], msg["encoding"], msg["step"]
    raw = np.frombuffer(msg["data"], np.uint8).reshape(h, step)
    if enc == "bgr8":
        return np.ascontiguousarray(raw[:, :3 * w].reshape(h, w, 3))
    if enc == "rgb8":
        return np.ascontiguousarray(raw[:, :3 * w].reshape(h, w, 3)[..., ::-1])
    if enc == "mono8":
        return np.ascontiguousarray(np.repeat(raw[:, :w, None], 3, axis=2))
    raise NotImplementedError(f"image encoding {enc}")


def read_sequence(path, image_topic="/pylon_camera/image_raw/compressed", altitude_topic="/drivers/altitude", altitude_field="altitude",
                  max_frames=None, use_header_stamp=True, relative_stamps=False, decode_workers=1):
    """Frames, stamps [s] and the latest altitude at each frame, in bag order - the node's view of its topics
    (it keeps the latest message of every topic, src/visual_Odometry.cpp:112-177).  decode_workers > 1 decodes the image
    payloads on that many host threads (Pillow releases the interpreter lock while it decodes: a 2560 x 2056 JPEG costs
    ~15 ms of one core, the front end takes a frame in well under a millisecond); the result is the same list, in order."""
    bag = BagReader(path)
    frames, stamps, alts = [], [], []
    last_alt = float("nan")
    pool = None
    if decode_workers > 1:
        from concurrent.futures import ThreadPoolExecutor

        pool = ThreadPoolExecutor(decode_workers)
    for topic, conn, t, raw in bag.messages([image_topic, altitude_topic]):
        msg = conn.codec.decode(raw)
        if topic == altitude_topic:
            last_alt = float(msg[altitude_field])
            continue
        frames.append(pool.submit(decode_image, msg, conn.type) if pool else decode_image(msg, conn.type))
        stamps.append(msg["header"]["stamp"] if use_header_stamp and "header" in msg else t)
        alts.append(last_alt)
        if max_frames is not None and len(frames) >= max_frames:
            break
    if pool:
        try:
            frames = [f.result() for f in frames]
        finally:
            pool.shutdown()
    if relative_stamps:
        # seconds since the first frame, from the integer (sec, nsec) pairs: exact differences (the float form of an epoch
        # stamp resolves only ~0.2 us), which is what the replay's dt needs
        ns = np.array([st.ns for st in stamps], np.int64)
        return frames, (ns - ns[0]) * 1e-9 if len(ns) else np.zeros(0), np.array(alts)
    return frames, np.array([float(st) for st in stamps]), np.array(alts)


def read_image_dir(path, fps=10.0, exts=(".png", ".jpg", ".jpeg", ".bmp", ".pgm", ".ppm")):
    """Image-directory source for sites without bags: files in name order, BGR frames, stamps at a nominal rate."""
    from PIL import Image

    names = sorted(n for n in os.listdir(path) if n.lower().endswith(exts))
    frames = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(path, n)).convert("RGB"))[..., ::-1]) for n in names]
    return frames, np.arange(len(frames)) / fps, names


if __name__ == "__main__":
    import json
    import sys

    print(json.dumps(BagReader(sys.argv[1]).info(), indent=1))
