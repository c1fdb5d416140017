"""Frame source for bag replay: what feeds the integrator in the reference's offline tool.

The reference's `kimera_semantics_rosbag` (kimera_semantics_ros/src/kimera_semantics_rosbag.cpp:83-141) reads a bag
into memory (rosbag_data_provider.cpp:83-193: depth / semantic / rgb images, one CameraInfo, every /tf and /tf_static
transform; the static camera -> base_link transform is kept aside), then for every depth image
  * CHECKs that the semantic (and rgb) image carries the same stamp                       (:95-110)
  * looks up  T_G_B = world <- base_link  at the image stamp in the tf buffer             (:124-129; tf interpolates)
  * composes  T_G_C = T_G_B * T_B_C  with the static camera mount                         (:130-133)
  * hands the back-projected cloud and T_G_C to the server                                (:134)
and logs an error and skips the frame when the lookup fails (:135-137).

This module provides the same pipeline without ROS:
  compose()          T_G_B * T_B_C in minkindr's arithmetic (float32: voxblox::Transformation is the float kind)
  TfBuffer           stamped transforms per (parent, child), lookup with tf's interpolation rule, no extrapolation
  synthetic_sequence the bench's trajectory as base_link poses + a fixed camera mount (the bag stand-in)
  read_rosbag        a reader for ROS1 bag files, format 2.0 (uncompressed and bz2 chunks), in pure Python:
                     sensor_msgs/Image, sensor_msgs/CameraInfo, tf2_msgs/TFMessage — the message types the
                     reference's provider instantiates
  replay()           the loop above, feeding ks_integrate_depth (the depth+label entry, row f-1)
`write_bag()` writes the same format (tests generate their fixture with it; no binary is committed).
"""
from __future__ import annotations

import bz2
import struct
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


# ---------------------------------------------------------------------------------------------------------------
# rigid transforms, [qw, qx, qy, qz, tx, ty, tz] float32 (the C ABI's T_G_C layout)
# ---------------------------------------------------------------------------------------------------------------
def quat_mul(a, b):
    """Hamilton product a * b, components (w, x, y, z), float32, Eigen's evaluation order."""
    aw, ax, ay, az = (F32(v) for v in a)
    bw, bx, by, bz = (F32(v) for v in b)
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx], dtype=F32)


def quat_rotate(q, v):
    """Eigen's Quaternion::_transformVector: v + w * (2 u x v) + u x (2 u x v), float32."""
    w = F32(q[0])
    u = np.asarray(q[1:4], dtype=F32)
    v = np.asarray(v, dtype=F32)
    uv = np.cross(u, v).astype(F32)
    uv = (uv + uv).astype(F32)
    return (v + w * uv + np.cross(u, uv).astype(F32)).astype(F32)


def compose(T_A_B, T_B_C):
    """T_A_C = T_A_B * T_B_C (kindr::minimal::QuatTransformation::operator*): q_A_C = q_A_B * q_B_C,
    t_A_C = t_A_B + q_A_B.rotate(t_B_C).  kimera_semantics_rosbag.cpp:130-133."""
    T_A_B = np.asarray(T_A_B, dtype=F32)
    T_B_C = np.asarray(T_B_C, dtype=F32)
    q = quat_mul(T_A_B[:4], T_B_C[:4])
    t = (T_A_B[4:7] + quat_rotate(T_A_B[:4], T_B_C[4:7])).astype(F32)
    return np.concatenate([q, t]).astype(F32)


def inverse(T):
    T = np.asarray(T, dtype=F32)
    qi = np.array([T[0], -T[1], -T[2], -T[3]], dtype=F32)
    return np.concatenate([qi, (-quat_rotate(qi, T[4:7])).astype(F32)]).astype(F32)


def _slerp(q0, q1, r):
    """tf::Quaternion::slerp (shortest arc; falls back to q0 when the quaternions coincide)."""
    q0 = np.asarray(q0, dtype=np.float64)
    q1 = np.asarray(q1, dtype=np.float64)
    d = float(np.dot(q0, q1))
    if d < 0.0:
        q1, d = -q1, -d
    d = min(1.0, d)
    theta = np.arccos(d)
    if theta < 1e-9:
        return q0.astype(F32)
    s = np.sin(theta)
    q = (np.sin((1.0 - r) * theta) * q0 + np.sin(r * theta) * q1) / s
    return (q / np.linalg.norm(q)).astype(F32)


class TfBuffer:
    """Stamped transforms parent <- child; lookup(parent, child, stamp) interpolates between the two neighbouring
    stamps like tf::TimeCache (linear translation, slerp rotation) and FAILS outside the covered interval, which is
    what makes the reference skip a frame ("Couldn't find tf for given pointcloud", rosbag.cpp:135-137).
    Stamps are integer nanoseconds.  Chains (world <- odom <- base_link) are followed parent by parent."""

    def __init__(self):
        self._edges = {}     # child -> (parent, [(stamp, T)])  sorted by stamp
        self._static = {}    # child -> (parent, T)

    def set_transform(self, stamp_ns: int, parent: str, child: str, T, static: bool = False):
        T = np.asarray(T, dtype=F32)
        if static:
            self._static[child] = (parent, T)
            return
        parent0, lst = self._edges.setdefault(child, (parent, []))
        if parent0 != parent:
            self._edges[child] = (parent, [])
            lst = self._edges[child][1]
        if lst and stamp_ns < lst[-1][0]:
            lst.append((int(stamp_ns), T))
            lst.sort(key=lambda e: e[0])
        else:
            lst.append((int(stamp_ns), T))

    def _edge(self, child, stamp_ns):
        if child in self._static:
            return self._static[child]
        if child not in self._edges:
            return None
        parent, lst = self._edges[child]
        stamps = [e[0] for e in lst]
        i = int(np.searchsorted(stamps, stamp_ns))
        if i < len(lst) and lst[i][0] == stamp_ns:
            return parent, lst[i][1]
        if i == 0 or i == len(lst):
            return None      # would be an extrapolation
        (s0, T0), (s1, T1) = lst[i - 1], lst[i]
        r = (stamp_ns - s0) / float(s1 - s0)
        t = (T0[4:7].astype(np.float64) * (1.0 - r) + T1[4:7].astype(np.float64) * r).astype(F32)
        return parent, np.concatenate([_slerp(T0[:4], T1[:4], r), t]).astype(F32)

    def lookup(self, target: str, source: str, stamp_ns: int):
        """T_target_source at stamp, or None."""
        T = np.array([1, 0, 0, 0, 0, 0, 0], dtype=F32)
        frame = source
        for _ in range(64):
            if frame == target:
                return T
            e = self._edge(frame, stamp_ns)
            if e is None:
                return None
            parent, T_p_f = e
            T = compose(T_p_f, T)
            frame = parent
        return None


# ---------------------------------------------------------------------------------------------------------------
# frames
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class SensorFrame:
    stamp_ns: int
    depth: np.ndarray                 # [H, W] float32 metres or uint16 millimetres
    semantic_rgba: np.ndarray | None  # [H, W, 4] colour-coded labels (what the bag holds)
    label_img: np.ndarray | None      # [H, W] uint8 labels when the source knows them directly
    K: tuple                          # fx, fy, cx, cy
    semantic_stamp_ns: int | None = None


@dataclass
class Sequence:
    frames: list = field(default_factory=list)
    tf: TfBuffer = field(default_factory=TfBuffer)
    T_B_C: np.ndarray = field(default_factory=lambda: np.array([1, 0, 0, 0, 0, 0, 0], dtype=F32))  # static camera mount
    world_frame: str = "world"
    base_link_frame: str = "base_link_gt"


def synthetic_sequence(n_frames: int, width=640, height=480, hfov_deg=90.0, scene="room", radius=1.5, dt_ns=33_333_333,
                       tf_rate_divisor=1, mount=(0.1, 0.0, 0.05)) -> Sequence:
    """The bench's trajectory ("kimera_semantics_demo.bag" stand-in) as a bag would hold it: base_link poses on /tf,
    a fixed camera mount on /tf_static, depth + colour-coded semantic images.  tf_rate_divisor > 1 publishes the pose
    only every k-th frame, so the stamps in between are interpolated by the buffer."""
    from . import synth
    sc = synth.make_scene(scene)
    seq = Sequence()
    T_B_C = np.array([1, 0, 0, 0, *mount], dtype=F32)       # camera ahead of / above base_link, same orientation
    seq.T_B_C = T_B_C
    T_C_B = inverse(T_B_C)
    colors = synth.default_label_colors()
    for k in range(n_frames):
        T_G_C = synth.trajectory_pose(k, radius=radius)
        f = synth.render_frame(sc, T_G_C, width, height, hfov_deg=hfov_deg, seed=k)
        stamp = 1_000_000_000 + k * dt_ns
        if k % tf_rate_divisor == 0 or k == n_frames - 1:
            seq.tf.set_transform(stamp, seq.world_frame, seq.base_link_frame, compose(T_G_C, T_C_B))
        seq.frames.append(SensorFrame(stamp, f.depth, np.ascontiguousarray(colors[f.label_img]), f.label_img, tuple(float(x) for x in f.K)))
    return seq


def replay(seq: Sequence, integrator, use_label_img=True, on_frame=None) -> dict:
    """kimera_semantics_rosbag.cpp:83-141 over a Sequence: per depth image the stamp CHECK, the tf lookup, the
    composition T_G_C = T_G_B * T_B_C and one ks_integrate_depth call.  Returns counters (frames integrated / skipped)."""
    done = skipped = 0
    for fr in seq.frames:
        if fr.semantic_stamp_ns is not None and fr.semantic_stamp_ns != fr.stamp_ns:
            raise ValueError(f"Depth and semantic image timestamps do not match: {fr.stamp_ns} vs {fr.semantic_stamp_ns}")  # CHECK_EQ, :95-99
        T_G_B = seq.tf.lookup(seq.world_frame, seq.base_link_frame, fr.stamp_ns)
        if T_G_B is None:
            skipped += 1      # "Couldn't find tf for given pointcloud..."
            continue
        T_G_C = compose(T_G_B, seq.T_B_C)
        if use_label_img and fr.label_img is not None:
            st = integrator.integrate_depth(T_G_C, fr.depth, fr.K, label_img=fr.label_img)
        else:
            st = integrator.integrate_depth(T_G_C, fr.depth, fr.K, rgba_img=fr.semantic_rgba)
        done += 1
        if on_frame:
            on_frame(fr, T_G_C, st)
    return {"integrated": done, "skipped_no_tf": skipped}


# ---------------------------------------------------------------------------------------------------------------
# ROS1 bag format 2.0 (http://wiki.ros.org/Bags/Format/2.0), the subset the reference's provider reads
# ---------------------------------------------------------------------------------------------------------------
OP_MSG, OP_BAG_HEADER, OP_INDEX, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 2, 3, 4, 5, 6, 7


def _read_record(buf, off):
    hlen, = struct.unpack_from("<I", buf, off)
    off += 4
    hdr, end = {}, off + hlen
    while off < end:
        flen, = struct.unpack_from("<I", buf, off)
        off += 4
        name, _, val = bytes(buf[off:off + flen]).partition(b"=")
        hdr[name.decode()] = val
        off += flen
    dlen, = struct.unpack_from("<I", buf, off)
    off += 4
    return hdr, memoryview(buf)[off:off + dlen], off + dlen


class _Cursor:
    def __init__(self, data):
        self.b, self.o = memoryview(data), 0

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def u32(self):
        v, = struct.unpack_from("<I", self.b, self.o)
        self.o += 4
        return v

    def f64s(self, n):
        v = np.frombuffer(self.b, "<f8", n, self.o).copy()
        self.o += 8 * n
        return v

    def string(self):
        n = self.u32()
        s = bytes(self.b[self.o:self.o + n]).decode()
        self.o += n
        return s

    def blob(self):
        n = self.u32()
        v = bytes(self.b[self.o:self.o + n])
        self.o += n
        return v

    def header(self):
        seq = self.u32()
        sec, nsec = self.u32(), self.u32()
        return seq, sec * 1_000_000_000 + nsec, self.string()


def parse_image(data):
    c = _Cursor(data)
    _, stamp, frame_id = c.header()
    h, w = c.u32(), c.u32()
    enc = c.string()
    big = c.u8()
    step = c.u32()
    raw = c.blob()
    bo = ">" if big else "<"
    if enc in ("32FC1",):
        img = np.frombuffer(raw, bo + "f4").reshape(h, step // 4)[:, :w].astype(np.float32)
    elif enc in ("16UC1", "mono16"):
        img = np.frombuffer(raw, bo + "u2").reshape(h, step // 2)[:, :w].astype(np.uint16)
    elif enc in ("rgb8", "bgr8"):
        img = np.frombuffer(raw, "u1").reshape(h, step)[:, :3 * w].reshape(h, w, 3)
        if enc == "bgr8":
            img = img[:, :, ::-1]
        img = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], axis=2)
    elif enc in ("rgba8", "bgra8"):
        img = np.frombuffer(raw, "u1").reshape(h, step)[:, :4 * w].reshape(h, w, 4)
        if enc == "bgra8":
            img = img[:, :, [2, 1, 0, 3]]
    elif enc in ("mono8", "8UC1"):
        img = np.frombuffer(raw, "u1").reshape(h, step)[:, :w]
    else:
        raise ValueError(f"image encoding {enc!r} not supported")
    return stamp, frame_id, np.ascontiguousarray(img)


def parse_camera_info(data):
    c = _Cursor(data)
    c.header()
    c.u32(), c.u32()
    c.string()
    c.f64s(c.u32())
    K = c.f64s(9)
    return (float(K[0]), float(K[4]), float(K[2]), float(K[5]))


def parse_tf_message(data):
    c = _Cursor(data)
    out = []
    for _ in range(c.u32()):
        _, stamp, parent = c.header()
        child = c.string()
        t = c.f64s(3)
        q = c.f64s(4)   # x, y, z, w
        out.append((stamp, parent.lstrip("/"), child.lstrip("/"), np.array([q[3], q[0], q[1], q[2], t[0], t[1], t[2]], dtype=F32)))
    return out


def iter_bag_messages(path):
    """(topic, datatype, receive time ns, raw message bytes) in file order."""
    buf = open(path, "rb").read()
    if not buf.startswith(b"#ROSBAG V2.0\n"):
        raise ValueError("not a ROS bag, format 2.0")
    conns = {}

    def connection(hdr, data):
        ch, c = {}, 0
        while c < len(data):
            flen, = struct.unpack_from("<I", data, c)
            name, _, val = bytes(data[c + 4:c + 4 + flen]).partition(b"=")
            ch[name.decode()] = val
            c += 4 + flen
        conns[struct.unpack("<I", hdr["conn"])[0]] = (hdr["topic"].decode(), ch.get("type", b"").decode())

    def walk(view):
        o = 0
        while o < len(view):
            hdr, data, o = _read_record(view, o)
            op = hdr["op"][0]
            if op == OP_CONNECTION:
                connection(hdr, data)
            elif op == OP_MSG:
                sec, nsec = struct.unpack("<II", hdr["time"])
                topic, dtype = conns[struct.unpack("<I", hdr["conn"])[0]]
                yield topic, dtype, sec * 1_000_000_000 + nsec, bytes(data)

    off = 13
    while off < len(buf):
        hdr, data, off = _read_record(buf, off)
        op = hdr["op"][0]
        if op == OP_CHUNK:
            comp = hdr.get("compression", b"none")
            if comp == b"none":
                inner = data
            elif comp == b"bz2":
                inner = memoryview(bz2.decompress(bytes(data)))
            else:
                raise ValueError(f"chunk compression {comp.decode()!r} not supported (none, bz2)")
            yield from walk(inner)
        elif op == OP_CONNECTION:   # (a recorded bag repeats its connection records after the chunks)
            connection(hdr, data)


def read_rosbag(path, depth_topic, semantic_topic, camera_info_topic, sensor_frame_id, base_link_frame_id="base_link_gt",
                world_frame_id="world") -> Sequence:
    """RosbagDataProvider::parseRosbag (rosbag_data_provider.cpp:83-193): images by topic, the first CameraInfo, every
    tf transform into the buffer — except sensor_frame <- base_link, which is the static camera mount."""
    seq = Sequence(world_frame=world_frame_id, base_link_frame=base_link_frame_id)
    depth, semantic, K = [], [], None
    for topic, dtype, _, raw in iter_bag_messages(path):
        if dtype == "sensor_msgs/Image":
            if topic == depth_topic:
                depth.append(parse_image(raw))
            elif topic == semantic_topic:
                semantic.append(parse_image(raw))
        elif dtype == "sensor_msgs/CameraInfo" and topic == camera_info_topic:
            K = parse_camera_info(raw)
        elif dtype in ("tf2_msgs/TFMessage", "tf/tfMessage"):
            for stamp, parent, child, T in parse_tf_message(raw):
                if child == sensor_frame_id and parent == base_link_frame_id:
                    seq.T_B_C = T
                else:
                    seq.tf.set_transform(stamp, parent, child, T, static=(topic == "/tf_static"))
    if not depth:
        raise ValueError("No depth images parsed from rosbag.")
    if len(depth) != len(semantic):
        raise ValueError("Unequal number of depth and semantic images.")
    if K is None:
        raise ValueError("no CameraInfo on " + camera_info_topic)
    for (sd, _, d), (ss, _, s) in zip(depth, semantic):
        seq.frames.append(SensorFrame(sd, d, s if s.ndim == 3 else None, s if s.ndim == 2 else None, K, semantic_stamp_ns=ss))
    return seq


# ---- writer (tests build their fixture with it) ----------------------------------------------------------------
def _field(name, val):
    b = name.encode() + b"=" + val
    return struct.pack("<I", len(b)) + b


def _record(hdr_fields, data):
    h = b"".join(_field(k, v) for k, v in hdr_fields)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def _ser_header(seq, stamp_ns, frame_id):
    f = frame_id.encode()
    return struct.pack("<III", seq, stamp_ns // 1_000_000_000, stamp_ns % 1_000_000_000) + struct.pack("<I", len(f)) + f


def ser_image(seq, stamp_ns, frame_id, img):
    if img.dtype == np.float32:
        enc, step = b"32FC1", 4 * img.shape[1]
    elif img.dtype == np.uint16:
        enc, step = b"16UC1", 2 * img.shape[1]
    elif img.ndim == 3 and img.shape[2] == 4:
        enc, step = b"rgba8", 4 * img.shape[1]
    else:
        enc, step = b"mono8", img.shape[1]
    raw = np.ascontiguousarray(img).tobytes()
    return (_ser_header(seq, stamp_ns, frame_id) + struct.pack("<II", img.shape[0], img.shape[1]) + struct.pack("<I", len(enc)) + enc +
            struct.pack("<BI", 0, step) + struct.pack("<I", len(raw)) + raw)


def ser_camera_info(stamp_ns, frame_id, w, h, K):
    fx, fy, cx, cy = K
    dm = b"plumb_bob"
    Km = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], "<f8")
    return (_ser_header(0, stamp_ns, frame_id) + struct.pack("<II", h, w) + struct.pack("<I", len(dm)) + dm + struct.pack("<I", 5) +
            np.zeros(5, "<f8").tobytes() + Km.tobytes() + np.eye(3, dtype="<f8").tobytes() +
            np.array([fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0], "<f8").tobytes() + struct.pack("<IIIIIIB", 0, 0, 0, 0, 0, 0, 0))


def ser_tf(transforms):
    out = struct.pack("<I", len(transforms))
    for stamp, parent, child, T in transforms:
        c = child.encode()
        out += _ser_header(0, stamp, parent) + struct.pack("<I", len(c)) + c
        out += np.array([T[4], T[5], T[6]], "<f8").tobytes() + np.array([T[1], T[2], T[3], T[0]], "<f8").tobytes()
    return out


def write_bag(path, messages, compression="none", chunk_messages=8):
    """messages: list of (topic, datatype, time_ns, serialized bytes).  Chunks of chunk_messages messages; the index and
    chunk-info records a player would need are omitted (a sequential reader does not use them)."""
    conn_of = {}
    body = [b"#ROSBAG V2.0\n"]
    hdr = _record([("op", bytes([OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 0)),
                   ("chunk_count", struct.pack("<I", 0))], b"")
    body.append(hdr)
    chunk = []

    def flush():
        if not chunk:
            return
        raw = b"".join(chunk)
        data = bz2.compress(raw) if compression == "bz2" else raw
        body.append(_record([("op", bytes([OP_CHUNK])), ("compression", compression.encode()), ("size", struct.pack("<I", len(raw)))], data))
        chunk.clear()

    n = 0
    for topic, dtype, t_ns, raw in messages:
        if topic not in conn_of:
            cid = len(conn_of)
            conn_of[topic] = cid
            ch = _field("topic", topic.encode()) + _field("type", dtype.encode()) + _field("md5sum", b"*") + _field("message_definition", b"")
            chunk.append(_record([("op", bytes([OP_CONNECTION])), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], ch))
        chunk.append(_record([("op", bytes([OP_MSG])), ("conn", struct.pack("<I", conn_of[topic])),
                              ("time", struct.pack("<II", t_ns // 1_000_000_000, t_ns % 1_000_000_000))], raw))
        n += 1
        if n % chunk_messages == 0:
            flush()
    flush()
    with open(path, "wb") as fh:
        fh.write(b"".join(body))


def sequence_to_messages(seq: Sequence, depth_topic="/depth", semantic_topic="/semantic", info_topic="/camera_info",
                         sensor_frame="left_cam"):
    """A Sequence as the messages of the reference's demo bag layout (depth, colour-coded semantic image, CameraInfo,
    /tf for the base_link pose, /tf_static for the camera mount)."""
    msgs = []
    first = seq.frames[0]
    h, w = first.depth.shape
    msgs.append(("/tf_static", "tf2_msgs/TFMessage", first.stamp_ns, ser_tf([(first.stamp_ns, seq.base_link_frame, sensor_frame, seq.T_B_C)])))
    msgs.append((info_topic, "sensor_msgs/CameraInfo", first.stamp_ns, ser_camera_info(first.stamp_ns, sensor_frame, w, h, first.K)))
    child = seq.base_link_frame
    tf_list = seq.tf._edges.get(child, (seq.world_frame, []))[1]
    events = [(s, "tf", T) for s, T in tf_list] + [(f.stamp_ns, "img", f) for f in seq.frames]
    events.sort(key=lambda e: (e[0], e[1] != "tf"))
    for k, (stamp, kind, obj) in enumerate(events):
        if kind == "tf":
            msgs.append(("/tf", "tf2_msgs/TFMessage", stamp, ser_tf([(stamp, seq.world_frame, child, obj)])))
        else:
            msgs.append((depth_topic, "sensor_msgs/Image", stamp, ser_image(k, stamp, sensor_frame, obj.depth)))
            sem = obj.semantic_rgba if obj.semantic_rgba is not None else obj.label_img
            msgs.append((semantic_topic, "sensor_msgs/Image", obj.semantic_stamp_ns or stamp, ser_image(k, obj.semantic_stamp_ns or stamp, sensor_frame, sem)))
    return msgs
