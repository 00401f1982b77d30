"""Host-side mirror of the reference's inference API (back2future.lua), over libb2f.so.

    back2future = require('back2future')                 from back2future_amd import back2future
    computeFlow = back2future.init('Ours-Soft-ft-KITTI') computeFlow = back2future.init('Ours-Soft-ft-KITTI')
    flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)  flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)

Same names, argument order, return order and error behaviour (a failing call raises, like
error() in Lua).  Images are 3 x H x W float arrays in [0,1] (what image.load returns);
flow is a 2 x H x W float64 array (raw network flow, NOT multiplied by 20, exactly like
back2future.lua:77-84), the masks are 1 x H x W uint8 arrays.  The Lua original keeps
`model` in a global (back2future.lua:113): one model per Lua state.  Here the model lives in
the returned closure, and `init` may be called several times.
"""
import ctypes as C

import numpy as np

from . import _lib

meanstd = {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}   # back2future.lua:33-36
occ_threshold = 0.6666                                                     # back2future.lua:40


def normalize(imgs):
    """M.normalize = TF.ColorNormalize(meanstd) (back2future.lua:42-45, transforms.lua:33-45)."""
    out = np.array(imgs, dtype=np.float32, copy=True)
    for c in range(out.shape[0]):
        out[c] = (out[c] + np.float32(-meanstd["mean"][c % 3])) / np.float32(meanstd["std"][c % 3])
    return out


class Model(object):
    """Owns a b2f_ctx (the `model` global of back2future.lua:113)."""

    def __init__(self, name="Ours-Soft-ft-KITTI", device=0, graph=None):
        """graph: createModelMulti options other than the shipped ones, e.g. "win=5,levels=4" (b2f_init_ex)."""
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.b2f_init_ex(name.encode() if name is not None else None, int(device),
                                 graph.encode() if graph else None, C.byref(h)))
        self._h = h
        self.name = name
        self.device = int(device)
        lv, win, pf, no, npar = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        _lib.check(L.b2f_info(h, C.byref(lv), C.byref(win), C.byref(pf), C.byref(no), C.byref(npar)))
        self.levels, self.win, self.past_flow = lv.value, win.value, bool(pf.value)
        self.n_outputs, self.n_params = no.value, npar.value

    def close(self):
        if getattr(self, "_h", None):
            if _lib is not None:            # None while the interpreter tears the module down
                _lib.lib().b2f_destroy(self._h)
            self._h = None

    __del__ = close

    # -- weights --
    def set_weights(self, flat):
        flat = _lib.f32(flat).ravel()
        _lib.check(_lib.lib().b2f_set_weights(self._h, _lib.fptr(flat), flat.size))
        self.past_flow = flat.size == _lib.lib().b2f_param_count(1)
        self.n_params = flat.size

    def get_weights(self):
        out = np.empty(self.n_params, np.float32)
        _lib.check(_lib.lib().b2f_get_weights(self._h, _lib.fptr(out), out.size))
        return out

    def weights_device_ptr(self):
        p, n = C.c_void_p(), C.c_longlong()
        _lib.check(_lib.lib().b2f_weights_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def commit_weights(self):
        _lib.check(_lib.lib().b2f_commit_weights(self._h))

    def set_option(self, key, value):
        _lib.check(_lib.lib().b2f_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int()
        _lib.check(_lib.lib().b2f_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def options(self, **kv):
        """Context manager: set options for the duration of a `with` block, then restore the previous values."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            old = {k: self.get_option(k) for k in kv}
            try:
                for k, v in kv.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return _cm()

    def synchronize(self):
        _lib.check(_lib.lib().b2f_synchronize(self._h))

    def profile_reset(self):
        _lib.check(_lib.lib().b2f_profile_reset(self._h))

    def profile_read(self):
        cap = 256
        names = C.create_string_buffer(32 * cap)
        ms = (C.c_double * cap)()
        cnt = (C.c_longlong * cap)()
        n = C.c_int()
        _lib.check(_lib.lib().b2f_profile_read(self._h, names, ms, cnt, cap, C.byref(n)))
        out = {}
        for i in range(n.value):
            nm = names.raw[32 * i:32 * i + 32].split(b"\0", 1)[0].decode()
            out[nm] = (ms[i], cnt[i])
        return out

    # -- the hot path --
    def computeFlow(self, im1, im2, im3):
        """computeFlow(im1, im2, im3) of back2future.lua:47-95."""
        im1, im2, im3 = _lib.f32(im1), _lib.f32(im2), _lib.f32(im3)
        assert im1.ndim == 3 and im1.shape[0] == 3 and im1.shape == im2.shape == im3.shape, \
            "expected three 3 x H x W images"
        _, H0, W0 = im1.shape
        flow = np.empty((2, H0, W0), np.float64)
        fwd = np.empty((1, H0, W0), np.uint8)
        bwd = np.empty((1, H0, W0), np.uint8)
        _lib.check(_lib.lib().b2f_compute_flow(
            self._h, _lib.fptr(im1), _lib.fptr(im2), _lib.fptr(im3), H0, W0,
            flow.ctypes.data_as(C.POINTER(C.c_double)), fwd.ctypes.data_as(C.POINTER(C.c_ubyte)),
            bwd.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return flow, fwd, bwd

    def computeFlowBatch(self, im1, im2, im3, out=None):
        """n independent triplets at once: inputs n x 3 x H x W.  The library pipelines sub-batches through
        pinned staging buffers; inputs / `out` = (flow f64 n x 2 x H x W, fwd u8 n x 1 x H x W, bwd) that already
        live in page-locked memory (e.g. views of torch pin_memory() tensors) are DMA'd in place instead.
        uint8 inputs (frames as decoded from 8-bit files, value = byte / 255) are uploaded as bytes."""
        as_bytes = all(np.asarray(a).dtype == np.uint8 for a in (im1, im2, im3))
        if as_bytes:
            im1, im2, im3 = (np.ascontiguousarray(a) for a in (im1, im2, im3))
        else:
            im1, im2, im3 = _lib.f32(im1), _lib.f32(im2), _lib.f32(im3)
        n, _, H0, W0 = im1.shape
        assert im1.shape == im2.shape == im3.shape and im1.shape[1] == 3, "expected three n x 3 x H x W arrays"
        if out is not None:
            flow, fwd, bwd = out
            assert flow.dtype == np.float64 and flow.shape == (n, 2, H0, W0) and flow.flags.c_contiguous
            for m in (fwd, bwd):
                assert m.dtype == np.uint8 and m.shape == (n, 1, H0, W0) and m.flags.c_contiguous
        else:
            flow = np.empty((n, 2, H0, W0), np.float64)
            fwd = np.empty((n, 1, H0, W0), np.uint8)
            bwd = np.empty((n, 1, H0, W0), np.uint8)
        outp = (flow.ctypes.data_as(C.POINTER(C.c_double)), fwd.ctypes.data_as(C.POINTER(C.c_ubyte)),
                bwd.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if as_bytes:
            u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
            _lib.check(_lib.lib().b2f_compute_flow_batch_u8(self._h, n, u8p(im1), u8p(im2), u8p(im3), H0, W0, *outp))
        else:
            _lib.check(_lib.lib().b2f_compute_flow_batch(self._h, n, _lib.fptr(im1), _lib.fptr(im2), _lib.fptr(im3), H0, W0, *outp))
        return flow, fwd, bwd

    def output_shapes(self, H, W):
        cap = 32
        ch, oh, ow = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        _lib.check(_lib.lib().b2f_output_shapes(self._h, H, W, ch, oh, ow, cap))
        return [(ch[i], oh[i], ow[i]) for i in range(self.n_outputs)]

    def forward(self, x):
        """model:forward(imgs) (back2future.lua:74): x is B x 9 x H x W, already normalized;
        returns the whole output table of pwc.lua:459-489 as a list of B x C x h x w arrays."""
        x = _lib.f32(x)
        B, nine, H, W = x.shape
        assert nine == 9
        outs = [np.empty((B, c, h, w), np.float32) for (c, h, w) in self.output_shapes(H, W)]
        ptrs = (_lib.c_float_p * len(outs))(*[_lib.fptr(o) for o in outs])
        _lib.check(_lib.lib().b2f_forward(self._h, _lib.fptr(x), B, H, W, ptrs, len(outs)))
        return outs

    def forward_device(self, d_in, B, H, W, d_flow=None, d_occ=None, d_est3=None, unit_input=False, stream=None):
        """model:forward on device pointers (ints); asynchronous on `stream`."""
        _lib.check(_lib.lib().b2f_forward_device(
            self._h, C.c_void_p(d_in), 1 if unit_input else 0, B, H, W,
            C.c_void_p(d_flow) if d_flow else None, C.c_void_p(d_occ) if d_occ else None,
            C.c_void_p(d_est3) if d_est3 else None, C.c_void_p(stream) if stream else None))


class MultiModel(object):
    """One process, several GPUs (b2f_init_multi): replaces nn.DataParallelTable of util.lua:27-48 for inference.  The
    weights are loaded once and broadcast to every GPU's replica (RCCL / peer copy) inside the library; computeFlowBatch
    splits the triplets contiguously over the GPUs."""

    TRANSPORT = {0: "single GPU", 1: "RCCL broadcast", 2: "hipMemcpyPeer"}

    def __init__(self, name="Ours-Soft-ft-KITTI", n_gpus=0, devices=None):
        L = _lib.lib()
        h = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices) if devices is not None else None
        _lib.check(L.b2f_init_multi(name.encode() if name is not None else None, int(n_gpus), dv, C.byref(h)))
        self._h = h
        n, tr = C.c_int(), C.c_int()
        devs = (C.c_int * 64)()
        _lib.check(L.b2f_multi_info(h, C.byref(n), devs, 64, C.byref(tr)))
        self.n_gpus, self.devices, self.transport = n.value, [devs[i] for i in range(n.value)], self.TRANSPORT[tr.value]

    def close(self):
        if getattr(self, "_h", None):
            if _lib is not None:
                _lib.lib().b2f_destroy_multi(self._h)
            self._h = None

    __del__ = close

    def weights_checksums(self):
        sums = (C.c_ulonglong * self.n_gpus)()
        _lib.check(_lib.lib().b2f_multi_weights_checksum(self._h, sums, self.n_gpus))
        return [int(v) for v in sums]

    def set_option(self, key, value):
        for i in range(self.n_gpus):
            _lib.check(_lib.lib().b2f_set_option(C.c_void_p(_lib.lib().b2f_multi_context(self._h, i)), key.encode(), int(value)))

    def computeFlowBatch(self, im1, im2, im3):
        as_bytes = all(np.asarray(a).dtype == np.uint8 for a in (im1, im2, im3))
        if as_bytes:
            im1, im2, im3 = (np.ascontiguousarray(a) for a in (im1, im2, im3))
        else:
            im1, im2, im3 = _lib.f32(im1), _lib.f32(im2), _lib.f32(im3)
        n, _, H0, W0 = im1.shape
        assert im1.shape == im2.shape == im3.shape and im1.shape[1] == 3, "expected three n x 3 x H x W arrays"
        flow = np.empty((n, 2, H0, W0), np.float64)
        fwd = np.empty((n, 1, H0, W0), np.uint8)
        bwd = np.empty((n, 1, H0, W0), np.uint8)
        outp = (flow.ctypes.data_as(C.POINTER(C.c_double)), fwd.ctypes.data_as(C.POINTER(C.c_ubyte)),
                bwd.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if as_bytes:
            u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
            _lib.check(_lib.lib().b2f_multi_compute_flow_batch_u8(self._h, n, u8p(im1), u8p(im2), u8p(im3), H0, W0, *outp))
        else:
            _lib.check(_lib.lib().b2f_multi_compute_flow_batch(self._h, n, _lib.fptr(im1), _lib.fptr(im2), _lib.fptr(im3), H0, W0, *outp))
        return flow, fwd, bwd


def shard_range(n, rank, world):
    """b2f_shard_range: [lo, hi) of `rank` when n triplets are split over `world` GPUs (util.lua:32)."""
    lo, hi = C.c_int(), C.c_int()
    _lib.check(_lib.lib().b2f_shard_range(int(n), int(rank), int(world), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def init(opt=None, device=0):
    """back2future.init(opt) (back2future.lua:97-129): returns the computeFlow closure.
    The closure carries the model as `.model`."""
    opt = opt or "Ours-Soft-ft-KITTI"
    model = Model(opt, device)

    def computeFlow(im1, im2, im3):
        return model.computeFlow(im1, im2, im3)

    computeFlow.model = model
    return computeFlow
