"""N4: rosbag 2.0 reader / generic message codec / image sources (visual_odometry_amd/bagio.py).
No ROS here: the container and wire format follow the published specifications and are exercised on bags written by the
module's own writer plus hand-assembled records, so compatibility with bags recorded by rosbag itself is unpinned."""
import io
import os
import struct

import numpy as np
import pytest

from visual_odometry_amd import bagio

SEP = "=" * 80
ALT_DEF = f"Header header\nfloat32 altitude\nbool validity\n{SEP}\nMSG: std_msgs/Header\n{bagio.HEADER_DEF}"
NAV_DEF = (f"# a nested, commented definition\nHeader header\nuint8 MODE_A=1   # constant\ngeometry_msgs/Vector3 position\nOrientation orientation\n"
           f"float64[3] cov\nint16[] samples\nstring[] names\ngeometry_msgs/Vector3[] trail\n"
           f"{SEP}\nMSG: std_msgs/Header\n{bagio.HEADER_DEF}{SEP}\nMSG: geometry_msgs/Vector3\nfloat64 x\nfloat64 y\nfloat64 z\n"
           f"{SEP}\nMSG: marta_msgs/Orientation\nfloat32 roll\nfloat32 pitch\nfloat32 yaw\n")


def test_codec_matches_the_ros1_wire_format_byte_for_byte():
    c = bagio.MsgCodec("marta_msgs/Altitude", ALT_DEF)
    raw = c.encode({"header": {"seq": 7, "stamp": 12.5, "frame_id": "alt"}, "altitude": 3.25, "validity": True})
    # uint32 seq | uint32 sec | uint32 nsec | uint32 len + "alt" | float32 | bool
    assert raw == struct.pack("<III", 7, 12, 500000000) + struct.pack("<I", 3) + b"alt" + struct.pack("<f", 3.25) + b"\x01"
    m = c.decode(raw)
    assert m["header"] == {"seq": 7, "stamp": 12.5, "frame_id": "alt"} and m["altitude"] == 3.25 and m["validity"] is True


def test_codec_nested_types_arrays_and_constants():
    c = bagio.MsgCodec("marta_msgs/NavStatus", NAV_DEF)
    msg = {"header": {"seq": 1, "stamp": 2.0, "frame_id": ""}, "position": {"x": 1.0, "y": -2.0, "z": 3.5},
           "orientation": {"roll": 0.1, "pitch": 0.2, "yaw": 0.3}, "cov": [1.0, 2.0, 3.0], "samples": [-1, 5, 300], "names": ["a", "bc"],
           "trail": [{"x": 0.0, "y": 1.0, "z": 2.0}, {"x": 3.0, "y": 4.0, "z": 5.0}]}
    raw = c.encode(msg)
    assert len(raw) == 12 + 4 + 24 + 12 + 24 + (4 + 6) + (4 + 5 + 6) + (4 + 48)
    m = c.decode(raw)
    assert m["position"] == msg["position"] and m["names"] == ["a", "bc"] and m["trail"] == msg["trail"]
    assert np.allclose(m["cov"], [1, 2, 3]) and list(m["samples"]) == [-1, 5, 300]
    assert abs(m["orientation"]["yaw"] - 0.3) < 1e-7
    assert "MODE_A" not in m


def _png(img_bgr):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(img_bgr[..., ::-1]).save(b, "PNG")
    return b.getvalue()


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_bag_round_trip_and_sequence_reader(tmp_path, compression):
    path = str(tmp_path / "t.bag")
    w = bagio.BagWriter(path, compression, chunk_bytes=30000)  # several chunks
    w.add_connection("/pylon_camera/image_raw/compressed", "sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    w.add_connection("/drivers/altitude", "marta_msgs/Altitude", ALT_DEF)
    w.add_connection("/other", "marta_msgs/NavStatus", NAV_DEF)
    ci = bagio.MsgCodec("sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    al = bagio.MsgCodec("marta_msgs/Altitude", ALT_DEF)
    imgs = []
    for i in range(7):
        img = np.random.default_rng(i).integers(0, 255, (60, 80, 3)).astype(np.uint8)
        imgs.append(img)
        if i != 3:  # frame 3 arrives without a new altitude: the node would still hold the previous one
            w.write("/drivers/altitude", 100 + 0.1 * i - 0.01, al.encode({"header": {"seq": i, "stamp": 100 + 0.1 * i - 0.01, "frame_id": ""},
                                                                          "altitude": 2.0 + i, "validity": True}))
        w.write("/pylon_camera/image_raw/compressed", 100 + 0.1 * i + 0.003,
                ci.encode({"header": {"seq": i, "stamp": 100 + 0.1 * i, "frame_id": "cam"}, "format": "png", "data": _png(img)}))
    w.close()
    # container level
    raw = open(path, "rb").read()
    assert raw.startswith(bagio.MAGIC) and len(raw) > 4096 + 13
    hlen = struct.unpack_from("<I", raw, 13)[0]
    dlen = struct.unpack_from("<I", raw, 13 + 4 + hlen)[0]
    assert 4 + hlen + 4 + dlen == 4096  # the bag header record is padded to 4096 bytes
    bag = bagio.BagReader(path)
    assert bag.conn_count == 3 and bag.chunk_count >= 2 and bag.index_pos > 4096
    info = bag.info()
    assert info["topics"]["/pylon_camera/image_raw/compressed"] == ("sensor_msgs/CompressedImage", 7)
    assert info["topics"]["/drivers/altitude"] == ("marta_msgs/Altitude", 6)
    # the node's view
    frames, stamps, alts = bagio.read_sequence(path)
    assert len(frames) == 7 and all(np.array_equal(a, b) for a, b in zip(frames, imgs))
    assert np.allclose(stamps, 100 + 0.1 * np.arange(7)) and np.allclose(alts, [2, 3, 4, 4, 6, 7, 8])
    f2, s2, _ = bagio.read_sequence(path, max_frames=3, use_header_stamp=False)
    assert len(f2) == 3 and np.allclose(s2, 100 + 0.1 * np.arange(3) + 0.003)
    # payloads decoded on host threads: the same frames in the same order
    f3, s3, a3 = bagio.read_sequence(path, decode_workers=3)
    assert len(f3) == 7 and all(np.array_equal(a, b) for a, b in zip(f3, imgs)) and np.array_equal(s3, stamps) and np.array_equal(a3, alts)


def test_hand_assembled_records():
    """A file put together field by field from the format specification (not through BagWriter)."""
    def field(k, v):
        return struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v

    def rec(fields, data):
        h = b"".join(field(k, v) for k, v in fields)
        return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data

    conn_hdr = field(b"topic", b"/x") + field(b"type", b"std_msgs/Float32") + field(b"md5sum", b"*") + field(b"message_definition", b"float32 data\n")
    conn = rec([(b"op", b"\x07"), (b"conn", struct.pack("<I", 0)), (b"topic", b"/x")], conn_hdr)
    msg = rec([(b"op", b"\x02"), (b"conn", struct.pack("<I", 0)), (b"time", struct.pack("<II", 5, 250000000))], struct.pack("<f", 1.5))
    chunk = rec([(b"op", b"\x05"), (b"compression", b"none"), (b"size", struct.pack("<I", len(conn + msg)))], conn + msg)
    bh = rec([(b"op", b"\x03"), (b"index_pos", struct.pack("<Q", 0)), (b"conn_count", struct.pack("<I", 1)), (b"chunk_count", struct.pack("<I", 1))], b"")
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".bag", delete=False) as f:
        f.write(bagio.MAGIC + bh + chunk)
    try:
        out = list(bagio.BagReader(f.name).messages())
        assert len(out) == 1
        topic, c, t, raw = out[0]
        assert topic == "/x" and c.type == "std_msgs/Float32" and t == 5.25 and c.codec.decode(raw) == {"data": 1.5}
    finally:
        os.unlink(f.name)


def test_image_decoding_and_directory_source(tmp_path):
    from PIL import Image

    img = np.zeros((40, 50, 3), np.uint8)
    img[..., 0], img[10:30, 10:40, 2] = 200, 180  # blue background, a red patch (BGR)
    b = io.BytesIO()
    Image.fromarray(img[..., ::-1]).save(b, "JPEG", quality=95)
    jpg = bagio.decode_image({"header": {}, "format": "jpeg", "data": b.getvalue()}, "sensor_msgs/CompressedImage")
    assert jpg.shape == img.shape and np.abs(jpg.astype(int) - img).mean() < 6  # lossy, but blue stays channel 0
    assert jpg[5, 5, 0] > 150 and jpg[20, 20, 2] > 130 and jpg[20, 20, 1] < 60
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)  # PNG payloads are lossless: every byte comes back, as BGR
    b = io.BytesIO()
    Image.fromarray(noise[..., ::-1]).save(b, "PNG")
    png = bagio.decode_image({"header": {}, "format": "png", "data": b.getvalue()}, "sensor_msgs/CompressedImage")
    assert np.array_equal(png, noise)
    raw = {"height": 40, "width": 50, "encoding": "rgb8", "is_bigendian": 0, "step": 152, "data": np.pad(img[..., ::-1].reshape(40, 150), ((0, 0), (0, 2))).tobytes()}
    assert np.array_equal(bagio.decode_image(raw, "sensor_msgs/Image"), img)
    mono = {"height": 40, "width": 50, "encoding": "mono8", "is_bigendian": 0, "step": 50, "data": img[..., 0].tobytes()}
    assert np.array_equal(bagio.decode_image(mono, "sensor_msgs/Image")[..., 1], img[..., 0])
    for i in (2, 0, 1):
        Image.fromarray(np.full((8, 8, 3), 10 * i, np.uint8)).save(tmp_path / f"f{i:03d}.png")
    frames, stamps, names = bagio.read_image_dir(str(tmp_path), fps=5.0)
    assert names == ["f000.png", "f001.png", "f002.png"] and [int(f[0, 0, 0]) for f in frames] == [0, 10, 20] and np.allclose(stamps, [0, 0.2, 0.4])


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.bag"
    p.write_bytes(b"#ROSBAG V1.2\n" + b"\0" * 64)
    with pytest.raises(ValueError):
        bagio.BagReader(str(p))
    with pytest.raises(KeyError):
        bagio.MsgCodec("a/B", "unknown_pkg/Thing t\n").decode(b"\0" * 8)


# ---- byte-level fixtures: every record type of the rosbag 2.0 specification, assembled from the specification's field tables
# (NOT through BagWriter), including what the sequential reader used to pass over: index data (op 4) and chunk info (op 6)
def _field(k, v):
    return struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v


def _rec(fields, data):
    h = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _conn_record(cid, topic, mtype, definition):
    hdr = _field(b"topic", topic) + _field(b"type", mtype) + _field(b"md5sum", b"0123456789abcdef0123456789abcdef") + \
        _field(b"message_definition", definition) + _field(b"callerid", b"/recorder") + _field(b"latching", b"0")  # optional fields
    return _rec([(b"conn", struct.pack("<I", cid)), (b"topic", topic), (b"op", b"\x07")], hdr)  # header fields in any order


def _lz4_frame(payload: bytes, blocks) -> bytes:
    """LZ4 frame around hand-made blocks: FLG = version 01, block-independent, content checksum; BD = 64 KiB blocks."""
    import xxhash

    flg, bd = 0b01100100, 0b01000000
    hc = (xxhash.xxh32(bytes([flg, bd]), seed=0).intdigest() >> 8) & 0xFF
    body = b"".join(struct.pack("<I", len(b) | (0x80000000 if stored else 0)) + b for b, stored in blocks)
    return struct.pack("<I", 0x184D2204) + bytes([flg, bd, hc]) + body + struct.pack("<I", 0) + \
        struct.pack("<I", xxhash.xxh32(payload, seed=0).intdigest())


def test_lz4_block_and_frame_decoding_against_hand_made_streams():
    # literals only
    assert bagio.lz4_block_decode(bytes([0x50]) + b"hello") == b"hello"
    # "abc" + match(offset 3, length 4 + 5 = 9, overlapping) + trailing literals "xyz!!" : token 0x35 / 0x50
    blk = bytes([0x35]) + b"abc" + struct.pack("<H", 3) + bytes([0x50]) + b"xyz!!"
    assert bagio.lz4_block_decode(blk) == b"abc" + b"abcabcabc" + b"xyz!!"
    # long literal run (15 + 255 + 30 = 300) and long match (19 + 255 + 1 = 275 zeros from offset 1)
    lit = bytes(range(256)) + bytes(44)
    blk = bytes([0xFF, 255, 30]) + lit + struct.pack("<H", 1) + bytes([255, 1]) + bytes([0x10]) + b"E"
    assert bagio.lz4_block_decode(blk) == lit + bytes([lit[-1]]) * 275 + b"E"
    with pytest.raises(ValueError):
        bagio.lz4_block_decode(bytes([0x04]) + struct.pack("<H", 9))  # offset before the start of the output
    payload = b"abc" + b"abcabcabc" + b"xyz!!" + b"STORED-BLOCK"
    frame = _lz4_frame(payload, [(bytes([0x35]) + b"abc" + struct.pack("<H", 3) + bytes([0x50]) + b"xyz!!", False), (b"STORED-BLOCK", True)])
    assert bagio.lz4_frame_decode(frame) == payload
    bad = bytearray(frame)
    bad[-1] ^= 1
    with pytest.raises(ValueError):
        bagio.lz4_frame_decode(bytes(bad))  # content checksum
    # the plain-Python XXH32 (header checksum, and every checksum when the xxhash module is absent) against the module
    import xxhash

    for n in (0, 1, 3, 4, 15, 16, 17, 63, 64, 1000):
        d = bytes((7 * i + n) & 255 for i in range(n))
        assert bagio._xxh32(d) == xxhash.xxh32(d, seed=0).intdigest(), n
    bad = bytearray(frame)
    bad[6] ^= 0x10
    with pytest.raises(ValueError, match="header checksum"):
        bagio.lz4_frame_decode(bytes(bad))
    for cut in (5, 6, 9, len(frame) - 9, len(frame) - 2):  # truncation is a ValueError, never IndexError / struct.error
        with pytest.raises(ValueError):
            bagio.lz4_frame_decode(frame[:cut])
    with pytest.raises(ValueError):
        bagio.lz4_block_decode(bytes([0x50]) + b"hel")  # literal run past the end
    with pytest.raises(ValueError):
        bagio.lz4_block_decode(bytes([0x35]) + b"abc" + b"\x03")  # match offset cut in half
    with pytest.raises(ValueError, match="larger than declared"):
        bagio.lz4_block_decode(bytes([0x50]) + b"hello", max_out=4)  # literals-only last sequence over the maximum


def _handmade_bag(compression):
    """Two chunks, two connections, index data after each chunk, index section with connection + chunk-info records."""
    f32 = b"float32 data\n"
    c0, c1 = _conn_record(0, b"/a", b"std_msgs/Float32", f32), _conn_record(1, b"/b", b"std_msgs/Float32", f32)

    def msg(cid, sec, nsec, val):
        return _rec([(b"op", b"\x02"), (b"time", struct.pack("<II", sec, nsec)), (b"conn", struct.pack("<I", cid))], struct.pack("<f", val))

    chunks = [[(c0, None), (msg(0, 10, 999999999, 1.0), (0, 10, 999999999)), (c1, None), (msg(1, 11, 5, 2.0), (1, 11, 5)),
               (msg(0, 11, 7, 3.0), (0, 11, 7))],
              [(msg(1, 20, 0, 4.0), (1, 20, 0)), (msg(1, 21, 500, 5.0), (1, 21, 500))]]
    body, infos = b"", []
    pos = len(bagio.MAGIC) + 4096
    for ch in chunks:
        raw, index = b"", {}
        for rec, key in ch:
            if key is not None:
                index.setdefault(key[0], []).append((key[1], key[2], len(raw)))
            raw += rec
        if compression == "none":
            data = raw
        elif compression == "bz2":
            import bz2

            data = bz2.compress(raw)
        else:  # lz4: one stored block + the rest as a literal-only compressed block
            half = len(raw) // 2
            lit = raw[half:]
            ll = len(lit)
            tok = bytes([0xF0]) + b"\xff" * ((ll - 15) // 255) + bytes([(ll - 15) % 255]) if ll >= 15 else bytes([ll << 4])
            data = _lz4_frame(raw, [(raw[:half], True), (tok + lit, False)])
        crec = _rec([(b"op", b"\x05"), (b"compression", compression.encode()), (b"size", struct.pack("<I", len(raw)))], data)
        irecs = b""
        for cid, ent in index.items():
            irecs += _rec([(b"op", b"\x04"), (b"ver", struct.pack("<I", 1)), (b"conn", struct.pack("<I", cid)), (b"count", struct.pack("<I", len(ent)))],
                          b"".join(struct.pack("<III", s_, n_, o_) for s_, n_, o_ in ent))
        times = [s_ * 10**9 + n_ for ent in index.values() for s_, n_, _ in ent]
        infos.append((pos + len(body), min(times), max(times), {cid: len(ent) for cid, ent in index.items()}))
        body += crec + irecs
    index_pos = pos + len(body)
    tail = c0 + c1
    for cpos, t0, t1, counts in infos:
        tail += _rec([(b"op", b"\x06"), (b"ver", struct.pack("<I", 1)), (b"chunk_pos", struct.pack("<Q", cpos)),
                      (b"start_time", struct.pack("<II", t0 // 10**9, t0 % 10**9)), (b"end_time", struct.pack("<II", t1 // 10**9, t1 % 10**9)),
                      (b"count", struct.pack("<I", len(counts)))], b"".join(struct.pack("<II", c, n) for c, n in counts.items()))
    bh_fields = [(b"op", b"\x03"), (b"index_pos", struct.pack("<Q", index_pos)), (b"conn_count", struct.pack("<I", 2)), (b"chunk_count", struct.pack("<I", 2))]
    h = b"".join(_field(k, v) for k, v in bh_fields)
    pad = 4096 - 4 - len(h) - 4
    bh = struct.pack("<I", len(h)) + h + struct.pack("<I", pad) + b" " * pad  # the bag header record is padded with spaces to 4096 bytes
    return bagio.MAGIC + bh + body + tail, infos, chunks


@pytest.mark.parametrize("compression", ["none", "bz2", "lz4"])
def test_every_record_type_from_the_format_specification(tmp_path, compression):
    blob, infos, chunks = _handmade_bag(compression)
    path = tmp_path / "hand.bag"
    path.write_bytes(blob)
    bag = bagio.BagReader(str(path))
    assert (bag.conn_count, bag.chunk_count) == (2, 2)
    seq = [(t, c.id, float(st), st.ns, c.codec.decode(raw)["data"]) for t, c, st, raw in bag.messages()]
    assert [(a, b, e) for a, b, _, _, e in seq] == [("/a", 0, 1.0), ("/b", 1, 2.0), ("/a", 0, 3.0), ("/b", 1, 4.0), ("/b", 1, 5.0)]
    assert [d for _, _, _, d, _ in seq] == [10_999_999_999, 11_000_000_005, 11_000_000_007, 20_000_000_000, 21_000_000_500]  # exact integer times
    # index section: connection records + chunk infos (op 6)
    bag2 = bagio.BagReader(str(path))
    assert bag2.read_index() == infos and sorted(bag2.connections) == [0, 1]
    assert bag2.connections[1].topic == "/b" and bag2.connections[1].md5sum == "0123456789abcdef0123456789abcdef"
    # index data records (op 4): offsets point at the message records inside the uncompressed chunk
    for (cpos, _, _, counts), ch in zip(infos, chunks):
        idx = bag2.chunk_index(cpos)
        assert {c: len(e) for c, e in idx.items()} == counts
        raw = b"".join(r for r, _ in ch)
        for cid, ent in idx.items():
            for t_ns, off in ent:
                fields, _ = bagio._read_header(memoryview(raw), off)
                assert fields["op"] == b"\x02" and struct.unpack("<I", fields["conn"])[0] == cid
                s_, n_ = struct.unpack("<II", fields["time"])
                assert s_ * 10**9 + n_ == t_ns
    # indexed access: by topic and by time range, skipping chunks the chunk infos rule out
    assert [c.codec.decode(r)["data"] for _, c, _, r in bag2.messages_indexed(["/a"])] == [1.0, 3.0]
    assert [c.codec.decode(r)["data"] for _, c, _, r in bag2.messages_indexed(None, start=11.0, end=20.5)] == [2.0, 3.0, 4.0]
    assert [c.codec.decode(r)["data"] for _, c, _, r in bag2.messages_indexed(["/b"], start=15.0)] == [4.0, 5.0]
    # an unclosed bag (index_pos = 0) still reads sequentially, but has no index section
    open_blob = bytearray(blob)
    ip = bytes(open_blob).index(b"index_pos=") + len(b"index_pos=")
    open_blob[ip : ip + 8] = struct.pack("<Q", 0)
    p2 = tmp_path / "open.bag"
    p2.write_bytes(bytes(open_blob))
    b3 = bagio.BagReader(str(p2))
    assert len(list(b3.messages())) == 5
    with pytest.raises(ValueError):
        b3.read_index()


def test_writer_chunk_infos_are_in_time_order_across_a_second_boundary(tmp_path):
    """The writer's chunk infos (used by messages_indexed) order times by (sec, nsec), not by the packed word."""
    path = str(tmp_path / "w.bag")
    w = bagio.BagWriter(path, "none", chunk_bytes=1 << 20)
    w.add_connection("/x", "std_msgs/Float32", "float32 data\n")
    for t, v in ((100.9, 1.0), (101.0001, 2.0), (101.5, 3.0)):
        w.write("/x", t, struct.pack("<f", v))
    w.close()
    bag = bagio.BagReader(path)
    (pos, t0, t1, counts), = bag.read_index()
    assert t0 == 100_900_000_000 and t1 == 101_500_000_000 and counts == {0: 3}
    assert [c.codec.decode(r)["data"] for _, c, _, r in bag.messages_indexed(["/x"], start=101.0)] == [2.0, 3.0]


def test_stamps_keep_integer_nanoseconds(tmp_path):
    """Header stamps at epoch scale: the float form resolves ~0.2 us, the (sec, nsec) pair is exact; read_sequence can return
    seconds since the first frame computed from the integers (what the replay's dt uses)."""
    c = bagio.MsgCodec("marta_msgs/Altitude", ALT_DEF)
    raw = struct.pack("<III", 1, 1_700_000_000, 123_456_789) + struct.pack("<I", 0) + struct.pack("<f", 1.0) + b"\x00"
    st = c.decode(raw)["header"]["stamp"]
    assert isinstance(st, float) and (st.sec, st.nsec) == (1_700_000_000, 123_456_789) and st.ns == 1_700_000_000_123_456_789
    assert c.encode(c.decode(raw)) == raw  # the pair survives a round trip bit for bit (the float alone would not)
    path = str(tmp_path / "s.bag")
    w = bagio.BagWriter(path)
    w.add_connection("/pylon_camera/image_raw/compressed", "sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    ci = bagio.MsgCodec("sensor_msgs/CompressedImage", bagio.COMPRESSED_IMAGE_DEF)
    img = np.zeros((8, 8, 3), np.uint8)
    for i, nsec in enumerate((999_999_999, 100_000_001, 200_000_003)):
        sec = 1_700_000_000 + (0 if i == 0 else 1)
        w.write("/pylon_camera/image_raw/compressed", 5.0 + i, ci.encode({"header": {"seq": i, "stamp": bagio.Stamp(sec, nsec), "frame_id": ""},
                                                                            "format": "png", "data": _png(img)}))
    w.close()
    _, rel, _ = bagio.read_sequence(path, relative_stamps=True)
    assert rel.tolist() == [0.0, 100_000_002 * 1e-9, 200_000_004 * 1e-9]
    _, ab, _ = bagio.read_sequence(path)
    assert abs((ab[1] - ab[0]) - 0.100000002) > 1e-9  # what the float form alone would give: off by more than a nanosecond
