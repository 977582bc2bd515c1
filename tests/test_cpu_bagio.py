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
