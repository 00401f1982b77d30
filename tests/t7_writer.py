"""Minimal Torch7 binary serializer + a builder of the serialized multi-frame PWC gModule
(tests only).

The reference's pretrained .t7 files are Dropbox links (README.md:49-52), none is in the tree and
Torch7 is not installed, so the .t7 reader (csrc/b2f_t7.cpp) is exercised against files written
here: same binary object format [3P torch7 File.lua] and the same object structure nngraph
produces for models/pwc.lua:87-508 (gModule.forwardnodes -> nngraph.Node{data={module=...},
children={...}}; nn.Sequential{modules={...}}; cudnn.SpatialConvolution{weight, bias,
nInputPlane, nOutputPlane, dW, ...}; shared storages for the siamese clones; optional
nn.DataParallelTable wrapper; torch.CudaTensor payloads).  This validates the reader against our
understanding of the format, not against a real file (DESIGN.md says so).
"""
import struct

import numpy as np

from back2future_amd import weights as W


class Storage(object):
    def __init__(self, cls, arr):
        self.cls, self.arr = cls, np.ascontiguousarray(arr)


class Tensor(object):
    def __init__(self, cls, storage, size, stride=None, offset=0):
        self.cls, self.storage, self.size = cls, storage, list(size)
        if stride is None:
            stride, s = [], 1
            for n in reversed(self.size):
                stride.insert(0, s)
                s *= n
        self.stride, self.offset = list(stride), offset


class TorchObj(object):
    def __init__(self, cls, **fields):
        self.cls, self.fields = cls, dict(fields)


class Function(object):
    """A serialized Lua function: tag 6 (TYPE_FUNCTION: size, dumped chunk, upvalues -- no reference index), 7
    (LEGACY_RECUR_FUNCTION) or 8 (TYPE_RECUR_FUNCTION): reference index, size, dumped chunk, upvalue table."""
    def __init__(self, tag, upvalues=None):
        self.tag, self.upvalues = tag, upvalues if upvalues is not None else {}


class Writer(object):
    def __init__(self, f, versioned=True):
        """versioned=False writes torch objects the pre-"V 1" way (class name string only)."""
        self.f, self.ids, self.next, self.versioned = f, {}, 1, versioned

    def header(self, cls):
        if self.versioned:
            self.string("V 1")
        self.string(cls)

    def i32(self, v): self.f.write(struct.pack("<i", int(v)))
    def i64(self, v): self.f.write(struct.pack("<q", int(v)))
    def f64(self, v): self.f.write(struct.pack("<d", float(v)))
    def string(self, s):
        b = s.encode()
        self.i32(len(b)); self.f.write(b)

    def ref(self, obj):
        """returns True if the object was written before (only its index is emitted)"""
        k = id(obj)
        if k in self.ids:
            self.i32(self.ids[k]); return True
        self.ids[k] = self.next; self.i32(self.next); self.next += 1
        return False

    def obj(self, o):
        if o is None:
            self.i32(0)
        elif isinstance(o, bool):
            self.i32(5); self.i32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.i32(1); self.f64(o)
        elif isinstance(o, str):
            self.i32(2); self.string(o)
        elif isinstance(o, (list, tuple)):
            self.i32(3)
            if self.ref(o): return
            self.i32(len(o))
            for i, v in enumerate(o):
                self.obj(i + 1); self.obj(v)
        elif isinstance(o, dict):
            self.i32(3)
            if self.ref(o): return
            self.i32(len(o))
            for k, v in o.items():
                self.obj(k); self.obj(v)
        elif isinstance(o, Storage):
            self.i32(4)
            if self.ref(o): return
            self.header(o.cls)
            self.i64(o.arr.size); self.f.write(o.arr.tobytes())
        elif isinstance(o, Tensor):
            self.i32(4)
            if self.ref(o): return
            self.header(o.cls)
            self.i32(len(o.size))
            for s in o.size: self.i64(s)
            for s in o.stride: self.i64(s)
            self.i64(o.offset + 1)
            self.obj(o.storage)
        elif isinstance(o, TorchObj):
            self.i32(4)
            if self.ref(o): return
            self.header(o.cls)
            self.obj(o.fields)
        elif isinstance(o, Function):
            self.i32(o.tag)
            if o.tag != 6 and self.ref(o): return
            self.string("\x1bLuaQ-not-a-real-chunk\x00\x01\x02")
            self.obj(o.upvalues)
        else:
            raise TypeError(type(o))


def _conv(cls_prefix, tcls, scls, w, b, stride, share=None):
    """cudnn/nn.SpatialConvolution with weight/bias (+ grads, like a checkpoint after clearState)."""
    co, ci = w.shape[0], w.shape[1]
    if share is None:
        ws, bs = Storage(scls, w.astype(np.float32)), Storage(scls, b.astype(np.float32))
        gws, gbs = Storage(scls, np.zeros(w.size, np.float32)), Storage(scls, np.zeros(b.size, np.float32))
    else:
        ws, bs, gws, gbs = share
    m = TorchObj(cls_prefix + ".SpatialConvolution",
                 nInputPlane=ci, nOutputPlane=co, kW=3, kH=3, dW=stride, dH=stride, padW=1, padH=1,
                 weight=Tensor(tcls, ws, (co, ci, 3, 3)), bias=Tensor(tcls, bs, (co,)),
                 gradWeight=Tensor(tcls, gws, (co, ci, 3, 3)), gradBias=Tensor(tcls, gbs, (co,)),
                 train=False)
    return m, (ws, bs, gws, gbs)


def build_model(flat, past_flow, cuda=True, cudnn=True, dpt=False, o=None):
    """The object tree torch.save(model) would produce for createModelMulti(opt) (pwc.lua:87-508) -- the shipped options, or with
    `o` (weights.graph_opts) another window / number of levels / pwc_skip (the other options at their defaults).  Only what
    the reader may look at is faithful: node/module classes, children links, Sequential contents, tensor sharing, MulConstant
    constants, the CostVolMulti nodes' win field, model fields."""
    o = o or W.SHIPPED
    LV, SK, WIN = o["levels"], o["skip"], o["win"]
    v = W.views(np.asarray(flat, np.float32), past_flow, o)
    tcls = "torch.CudaTensor" if cuda else "torch.FloatTensor"
    scls = "torch.CudaStorage" if cuda else "torch.FloatStorage"
    cp = "cudnn" if cudnn else "nn"
    nodes = []

    def node(module, *parents):
        n = TorchObj("nngraph.Node", data={"module": module, "mapindex": [p.fields["data"] for p in parents]},
                     children=[], visited=False, id=len(nodes) + 1)
        nodes.append(n)
        for p in parents:
            p.fields["children"].append(n)
        return n

    def seq(mods):
        return TorchObj("nn.Sequential", modules=list(mods), train=False)

    lrelu = lambda: TorchObj("nn.LeakyReLU", negval=0.2, inplace=True)

    inp = node(TorchObj("nn.Identity"))
    Is = {f: node(TorchObj("nn.Narrow", dimension=2, index=(f - 1) * 3 + 1, length=3), inp) for f in (1, 2, 3)}
    ds = {}
    for f in (1, 3):
        ds[f] = {1: Is[f]}
        for l in range(2, LV - SK + 1):
            ds[f][l] = node(TorchObj(cp + ".SpatialAveragePooling", kW=2, kH=2, dW=2, dH=2), ds[f][l - 1])
    # siamese feature towers: frames 2 and 3 are clones sharing the storages of frame 1 (pwc.lua:187-195)
    cs, shares = {}, {}
    for f in (1, 2, 3):
        cs[f] = {1: Is[f]}
        for l in range(1 if SK == 0 else 2, LV + 1):       # pwc_skip = 0: convUnit(3, 16, 1) on level 1 (pwc.lua:171-173)
            c1, s1 = _conv(cp, tcls, scls, v["feat%d.conv1.w" % l], v["feat%d.conv1.b" % l], 1 if l == 1 else 2, shares.get((l, 1)))
            c2, s2 = _conv(cp, tcls, scls, v["feat%d.conv2.w" % l], v["feat%d.conv2.b" % l], 1, shares.get((l, 2)))
            shares[(l, 1)], shares[(l, 2)] = s1, s2
            cs[f][l] = node(seq([c1, lrelu(), c2, lrelu()]), Is[f] if l == 1 else cs[f][l - 1])

    def decoder(l, kind):
        mods = []
        for i in range(1, 7):
            c, _ = _conv(cp, tcls, scls, v["l%d.%s.conv%d.w" % (l, kind, i)], v["l%d.%s.conv%d.b" % (l, kind, i)], 1)
            mods.append(c)
            if i < 6:
                mods.append(lrelu())
        return seq(mods)

    def warp(I, F):   # warpingUnit, pwc.lua:68-73
        a = node(TorchObj("nn.Transpose", permutations=[[2, 3], [3, 4]]), I)
        b = node(TorchObj("nn.Transpose", permutations=[[2, 3], [3, 4]]), F)
        s = node(TorchObj("nn.BilinearSamplerBHWD"), a, b)
        return node(TorchObj("nn.Transpose", permutations=[[3, 4], [2, 3]]), s)

    up = lambda p: node(TorchObj("nn.SpatialUpSamplingBilinear", scale_factor=2.0), p)
    nn2 = lambda p: node(TorchObj("nn.SpatialUpSamplingNearest", scale_factor=2.0), p)
    mulc = lambda p, k: node(TorchObj("nn.MulConstant", constant_scalar=float(k), inplace=False), p)
    ws = {1: {}, 3: {}}
    ufs, ubfs, outs = {}, {}, {}
    for l in range(LV, SK, -1):
        src = cs if l == LV else ws
        cvf = node(TorchObj("nn.CostVolMulti", win=WIN, fwd=True, verbose=False), cs[2][l], src[3][l])
        cvb = node(TorchObj("nn.CostVolMulti", win=WIN, fwd=False, verbose=False), cs[2][l], src[1][l])
        cv = node(TorchObj("nn.JoinTable", dimension=2), cvf, cvb)
        oin = [cv, cs[2][l]] + ([ufs[l + 1]] if l != LV else [])
        occ = node(TorchObj(cp + ".SpatialSoftMax"), node(decoder(l, "occ"), node(TorchObj("nn.JoinTable", dimension=2), *oin)))
        skip_occ = occ
        for _ in range(SK):
            skip_occ = nn2(skip_occ)
        if l == LV:
            fs = node(decoder(l, "flow"), cv)
            bfs = node(decoder(l, "past"), cv) if past_flow else None
        else:
            fs = node(decoder(l, "flow"), node(TorchObj("nn.JoinTable", dimension=2), cv, cs[2][l], ufs[l + 1]))
            bfs = node(decoder(l, "past"), node(TorchObj("nn.JoinTable", dimension=2), cv, cs[2][l], ubfs[l + 1])) if past_flow else None
        ufs[l] = up(fs) if (SK > 0 or l > 1) else None        # pwc.lua:359
        skip_u = ufs[l] if SK > 0 else fs                    # pwc.lua:423-429,462-466
        for _ in range(SK - 1):
            skip_u = up(skip_u)
        if past_flow:
            ubfs[l] = up(bfs) if (SK > 0 or l > 1) else None
            skip_ub = ubfs[l] if SK > 0 else bfs
            for _ in range(SK - 1):
                skip_ub = up(skip_ub)
        iws = {}
        for f in (1, 3):
            if l > SK + 1:
                ws[f][l - 1] = warp(cs[f][l - 1], mulc(ufs[l], 20.0 * (f - 2) / 2 ** (l - 2)))
            tmp = skip_ub if (past_flow and f < 2) else skip_u
            iws[f] = warp(ds[f][l - SK], mulc(tmp, 20.0 * (f - 2) / 2 ** (l - SK - 1)))
        outs[l] = [skip_u] + ([skip_ub] if past_flow else []) + [skip_occ, iws[1], iws[3]]
    out_nodes = [n for l in range(SK + 1, LV + 1) for n in outs[l]]
    outnode = node(TorchObj("nn.Identity"), *out_nodes)
    modules = [n.fields["data"]["module"] for n in nodes]
    g = TorchObj("nn.gModule", forwardnodes=nodes, modules=modules, outnode=outnode, innode=inp,
                 nInputs=1, verbose=False, train=False, past_flow=bool(past_flow),
                 flow_scale=[20.0 / 2 ** (l - SK - 1) for l in range(LV, SK, -1)])
    if dpt:
        g = TorchObj("nn.DataParallelTable", modules=[g], gpuAssignments=[1], dimension=1, train=False)
    return g


def save(path, flat, past_flow, **kw):
    with open(path, "wb") as f:
        Writer(f).obj(build_model(flat, past_flow, **kw))


def build_model_legacy(flat, past_flow, replicas=2, o=None):
    """A structurally different serialization of the same network, as an older / differently trained checkpoint could
    look: plain nn.* classes on torch.FloatTensor, ALL parameters as views (offset + strides) into ONE flat storage (what
    model:getParameters() leaves behind), convolution weights as 2-D Co x (Ci*9) views (SpatialConvolutionMM-style),
    the bias of every other conv as a strided view, double-precision MulConstant-irrelevant extras, unknown fields,
    serialized functions of all three tags, forward nodes in reverse order, and an nn.DataParallelTable with `replicas`
    gModules (back2future.lua:114-116 takes the first)."""
    g = build_model(flat, past_flow, cuda=False, cudnn=False, dpt=False, o=o)
    flatv = np.asarray(flat, np.float32)
    convs = []
    seen = set()

    def walk(o):
        if id(o) in seen:
            return
        seen.add(id(o))
        if isinstance(o, TorchObj):
            if o.cls.endswith("SpatialConvolution"):
                convs.append(o)
            walk(o.fields)
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
    walk(g)
    # one big storage: every distinct weight / bias array once (siamese clones keep sharing), biases interleaved with
    # a gap so that their views are strided
    pieces, where, total = [], {}, 3
    for c in convs:
        for name in ("weight", "bias"):
            st = c.fields[name].storage
            if id(st) not in where:
                strided = name == "bias" and (len(where) % 4 == 1)
                n = st.arr.size * (2 if strided else 1)
                where[id(st)] = (total, strided)
                pieces.append((total, st.arr, strided))
                total += n + 5
    big = np.full(total, np.float32(-777.0), np.float32)
    for off, arr, strided in pieces:
        if strided:
            big[off:off + 2 * arr.size:2] = arr.ravel()
        else:
            big[off:off + arr.size] = arr.ravel()
    store = Storage("torch.FloatStorage", big)
    for i, c in enumerate(convs):
        co, ci = int(c.fields["nOutputPlane"]), int(c.fields["nInputPlane"])
        woff, _ = where[id(c.fields["weight"].storage)]
        boff, bstr = where[id(c.fields["bias"].storage)]
        c.fields["weight"] = Tensor("torch.FloatTensor", store, (co, ci * 9), offset=woff)           # 2-D view
        c.fields["bias"] = Tensor("torch.FloatTensor", store, (co,), stride=[2 if bstr else 1], offset=boff)
        del c.fields["gradWeight"], c.fields["gradBias"]
        c.fields["finput"] = Tensor("torch.FloatTensor", Storage("torch.FloatStorage", np.zeros(0, np.float32)), ())
        c.fields["_type"] = "torch.FloatTensor"
        c.fields["accGradParameters"] = Function(6 if i % 3 == 0 else (7 if i % 3 == 1 else 8), {"n": i, "t": [1, 2, {"deep": True}]})
        c.fields[7] = {"numeric key": 1.5}
    g.fields["forwardnodes"] = list(reversed(g.fields["forwardnodes"]))
    g.fields["modules"] = list(reversed(g.fields["modules"]))
    g.fields["fg"] = TorchObj("graph.Graph", nodes=g.fields["forwardnodes"], edges=[], name="fg")
    g.fields["type"] = Function(8)
    second = TorchObj("nn.gModule", forwardnodes=[], modules=[], note="second replica: never read")
    dpt = TorchObj("nn.DataParallelTable", modules=[g] + [second] * (replicas - 1), gpuAssignments=list(range(1, replicas + 1)),
                   dimension=1, flattenedParams=[Tensor("torch.FloatTensor", store, (big.size,))], impl=Function(6))
    return dpt


def save_legacy(path, flat, past_flow, versioned=False, **kw):
    with open(path, "wb") as f:
        Writer(f, versioned=versioned).obj(build_model_legacy(flat, past_flow, **kw))
