"""numpy restatement of the tile-merge rule of ks_merge_tiles_device (the checker for the
multi-GPU reduce; f32, same operation order, no FMA) and a numpy-backed tile store so the
exchange protocol of kimera_semantics_amd.parallel can run on CPU over gloo."""
import numpy as np

PRIOR_INIT = np.float32(-0.60205999132)
TILE_WORDS = 16384


def _round_half_away(x):
    return np.floor(x + np.float32(0.5))


def merge_records(dst: np.ndarray, src: np.ndarray, max_weight=10000.0, color_mode=1, label_rgba=None):
    """dst, src: [..., 32] uint32 voxel records (dword 0 dist, 1 weight, 2 rgba, 3 label, 4..24 priors).
    Merges src into dst in place, following k_merge_tiles."""
    d = dst.reshape(-1, 32)
    s = src.reshape(-1, 32)
    touched = s[:, 3] != 255
    ad, aw = s[:, 0].view(np.float32), s[:, 1].view(np.float32)
    bd, bw = d[:, 0].view(np.float32).copy(), d[:, 1].view(np.float32).copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        cw = (aw + bw).astype(np.float32)
        upd = touched & (cw > 0)
        nd = ((ad * aw).astype(np.float32) + (bd * bw).astype(np.float32)).astype(np.float32) / cw
    nd = nd.astype(np.float32)
    if color_mode == 0:
        w1 = (aw / cw).astype(np.float32)
        w2 = (bw / cw).astype(np.float32)
        ca = s[:, 2:3].view(np.uint8).astype(np.float32)
        cb = d[:, 2:3].view(np.uint8).astype(np.float32)
        blended = _round_half_away((ca * w1[:, None]).astype(np.float32) + (cb * w2[:, None]).astype(np.float32))
        newc = np.ascontiguousarray(blended.astype(np.uint8)).view(np.uint32)[:, 0]
        d[:, 2] = np.where(upd, newc, d[:, 2])
    d[:, 0] = np.where(upd, nd.view(np.uint32), d[:, 0])
    d[:, 1] = np.where(upd, np.minimum(np.float32(max_weight), cw).astype(np.float32).view(np.uint32), d[:, 1])
    pa = s[:, 4:25].view(np.float32)
    pb = d[:, 4:25].view(np.float32)
    pn = (pb + (pa - PRIOR_INIT).astype(np.float32)).astype(np.float32)
    d[:, 4:25] = np.where(touched[:, None], pn.view(np.uint32), d[:, 4:25])
    best = np.argmax(d[:, 4:25].view(np.float32), axis=1).astype(np.uint32)
    d[:, 3] = np.where(touched, best, d[:, 3])
    if color_mode == 1:
        lut = np.ascontiguousarray(np.asarray(label_rgba, dtype=np.uint8).reshape(256, 4)).view(np.uint32)[:, 0]
        d[:, 2] = np.where(touched, lut[np.minimum(best, 255)], d[:, 2])
    return dst


def empty_tile():
    t = np.zeros((512, 32), dtype=np.uint32)
    t[:, 3] = 255
    t[:, 4:25] = PRIOR_INIT.view(np.uint32)
    return t


class NumpyTileStore:
    """key -> [512, 32] uint32 records; payload tensors are torch CPU int32 [n, 16384]."""

    def __init__(self, label_rgba, color_mode=1, max_weight=10000.0):
        self.tiles = {}
        self.order = []
        self.label_rgba, self.color_mode, self.max_weight = label_rgba, color_mode, max_weight

    def add(self, key, rec):
        if key not in self.tiles:
            self.order.append(key)
        self.tiles[key] = rec

    def tile_keys(self):
        return np.array(self.order, dtype=np.uint64)

    def export(self, slots):
        import torch
        out = np.zeros((len(slots), TILE_WORDS), dtype=np.int32)
        for i, s in enumerate(slots):
            out[i] = self.tiles[self.order[int(s)]].reshape(-1).view(np.int32)
        return torch.from_numpy(out)

    def empty(self, n):
        import torch
        return torch.empty((n, TILE_WORDS), dtype=torch.int32)

    def reset(self, slots):
        for s in slots:
            self.tiles[self.order[int(s)]] = empty_tile()

    def merge(self, keys, payload):
        p = payload.numpy().view(np.uint32).reshape(len(keys), 512, 32)
        for k, rec in zip(keys.tolist(), p):
            if k not in self.tiles:
                self.add(k, empty_tile())
            merge_records(self.tiles[k], rec, self.max_weight, self.color_mode, self.label_rgba)
