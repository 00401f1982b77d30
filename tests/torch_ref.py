"""Independent PyTorch-CPU witness of the ops the oracle restates (tests only).

PyTorch descends from the same THNN kernels as Torch7's nn, so it is a good but
independent check of the [3P] semantics.  It exists only in this
container; nothing here travels to the GPU box at run time except as committed
golden vectors (tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

FEAT = [0, 3, 16, 32, 64, 96, 128, 192]


def costvol_lua(ref, frame, win, fwd):
    """Literal transcription of the slicing loops of models/CostVolMulti.lua:62-100
    (two input frames) with torch slices."""
    B, N, h, w = ref.shape
    n = (win - 1) // 2
    out = torch.zeros(B, win * win, h, w, dtype=ref.dtype)
    i = 0
    for q_x_ in range(-n, n + 1):
        for q_y_ in range(-n, n + 1):
            q_x, q_y = q_x_, q_y_
            if not fwd:
                q_x, q_y = -q_x, -q_y
            # 1-based inclusive Lua ranges -> python slices
            if q_x < 0:
                qx, px = (1, w + q_x), (1 - q_x, w)
            else:
                qx, px = (1 + q_x, w), (1, w - q_x)
            if q_y < 0:
                qy, py = (1, h + q_y), (1 - q_y, h)
            else:
                qy, py = (1 + q_y, h), (1, h - q_y)
            if qx[1] >= qx[0] and qy[1] >= qy[0]:
                cost = ref[:, :, qy[0] - 1:qy[1], qx[0] - 1:qx[1]] * frame[:, :, py[0] - 1:py[1], px[0] - 1:px[1]]
                out[:, i, qy[0] - 1:qy[1], qx[0] - 1:qx[1]] += cost.sum(1)
            i += 1
    return out / N


def warp_grid_sample(img_bchw, flow_b2hw):
    """CUDA sampler semantics == grid_sample(border, align_corners=True) with
    g = 2 (x+u)/(w-1) - 1 (verified in the survey); needs h, w > 1."""
    B, C, h, w = img_bchw.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    gx = 2 * (xs[None] + flow_b2hw[:, 0].double()) / (w - 1) - 1
    gy = 2 * (ys[None] + flow_b2hw[:, 1].double()) / (h - 1) - 1
    grid = torch.stack([gx, gy], -1)
    return F.grid_sample(img_bchw.double(), grid, mode="bilinear", padding_mode="border", align_corners=True)


def warp_gather(img_bchw, flow_b2hw):
    """Same function written with explicit gathers in float64 (works for h or w == 1)."""
    B, C, h, w = img_bchw.shape
    img = img_bchw.double()
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    xc = (xs[None] + flow_b2hw[:, 0].double()).clamp(0, w - 1)
    yc = (ys[None] + flow_b2hw[:, 1].double()).clamp(0, h - 1)
    x0 = xc.floor(); y0 = yc.floor()
    wx = 1 - (xc - x0); wy = 1 - (yc - y0)
    x0 = x0.long(); y0 = y0.long()
    x1 = (x0 + 1).clamp(max=w - 1); y1 = (y0 + 1).clamp(max=h - 1)  # weight is 0 when clamped

    def g(yy, xx):
        idx = (yy * w + xx).view(B, 1, -1).expand(B, C, -1)
        return img.view(B, C, -1).gather(2, idx).view(B, C, h, w)

    wx = wx[:, None]; wy = wy[:, None]
    return wx * wy * g(y0, x0) + (1 - wx) * wy * g(y0, x1) + wx * (1 - wy) * g(y1, x0) + (1 - wx) * (1 - wy) * g(y1, x1)


def _conv(x, w, b, stride=1, leaky=True):
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    return F.leaky_relu(y, 0.2) if leaky else y


def pwc_forward(x, wv, past_flow, dtype=torch.float64, o=None):
    """models/pwc.lua createModelMulti(opt), in torch ops; o = option dict (back2future_amd.weights.graph_opts; default:
    the shipped opts).  x: B x 9 x H x W numpy, wv: dict name->numpy (weights.views).  Returns the output table as a
    list of numpy arrays plus a dict of intermediates."""
    from back2future_amd import weights as Wt
    o = o or Wt.SHIPPED
    L, LST, win = o["levels"], o["skip"] + 1, o["win"]
    siam, skip0 = o.get("siamese", 1), o["skip"] == 0
    ff = float(o["flownet_factor"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    x = t(x)
    P = {k: t(v) for k, v in wv.items()}
    Is = {f: x[:, 3 * (f - 1):3 * f] for f in (1, 2, 3)}
    ds = {}
    for f in (1, 3):
        ds[f] = {1: Is[f]}
        for k in range(2, L - LST + 2):
            ds[f][k] = F.avg_pool2d(ds[f][k - 1], 2)
    frames = (2, 3) if o["two_frame"] else (1, 2, 3)
    cs = {}
    for f in frames:
        cs[f] = {1: Is[f]}
        if skip0 and siam:                                 # pwc.lua:171-177,202-203: convUnit(3, 16, 1) on the image
            a = _conv(Is[f], P["feat1.conv1.w"], P["feat1.conv1.b"], 1)
            cs[f][1] = _conv(a, P["feat1.conv2.w"], P["feat1.conv2.b"], 1)
        for l in range(2, L + 1):
            if not siam:                                   # pwc.lua:182: nn.SpatialAveragePooling(2,2,2,2) instead of the convUnit
                cs[f][l] = F.avg_pool2d(cs[f][l - 1], 2)
                continue
            a = _conv(cs[f][l - 1], P["feat%d.conv1.w" % l], P["feat%d.conv1.b" % l], 2)
            cs[f][l] = _conv(a, P["feat%d.conv2.w" % l], P["feat%d.conv2.b" % l], 1)

    def dec(inp, l, kind):
        y = inp
        for i in range(1, 7):
            y = _conv(y, P["l%d.%s.conv%d.w" % (l, kind, i)], P["l%d.%s.conv%d.b" % (l, kind, i)], 1, leaky=i < 6)
        return y

    up = lambda a: F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
    nn2 = lambda a: F.interpolate(a, scale_factor=2, mode="nearest")
    ws = {1: {}, 3: {}}
    fs, bfs, ufs, ubfs, sk_u, sk_ub, occs, uoccs, sk_o, iws = {}, {}, {}, {}, {}, {}, {}, {}, {}, {1: {}, 3: {}}
    inter = {}
    for l in range(L, LST - 1, -1):
        src = cs if l == L else ws
        cvf = costvol_lua(cs[2][l], src[3][l], win, True)
        if not o["two_frame"]:
            cvb = costvol_lua(cs[2][l], src[1][l], win, False)
            cv_occ = torch.cat([cvf, cvb], 1)
            cv_flow = (cvf + cvb) if o["sum_cvs"] else cv_occ
        else:
            cv_occ = cv_flow = cvf
        inter["cv%d" % l] = cv_occ
        oin = [cv_occ, cs[2][l]] + ([cs[3][l]] if o["two_frame"] else [])
        if l != L:
            oin.append(ufs[l + 1])
            if o["occ_input"]:
                oin.append(uoccs[l + 1])
        occs[l] = F.softmax(dec(torch.cat(oin, 1), l, "occ"), dim=1)
        uoccs[l] = nn2(occs[l])
        sk_o[l] = uoccs[l]
        for _ in range(2, LST):
            sk_o[l] = nn2(sk_o[l])
        if l == L:
            fs[l] = dec(cv_flow, l, "flow")
            if past_flow:
                bfs[l] = dec(cv_flow, l, "past")
        else:
            fs[l] = dec(torch.cat([cv_flow, cs[2][l], ufs[l + 1]], 1), l, "flow")
            if past_flow:
                bfs[l] = dec(torch.cat([cv_flow, cs[2][l], ubfs[l + 1]], 1), l, "past")
            if o["residual"]:
                fs[l] = fs[l] + ufs[l + 1]
                if past_flow:
                    bfs[l] = bfs[l] + ubfs[l + 1]
        inter["fs%d" % l] = fs[l]
        mul = 2.0 if o["rescale_flow"] else 1.0
        ufs[l] = up(fs[l]) * mul
        sk_u[l] = ufs[l]
        for _ in range(2, LST):
            sk_u[l] = up(sk_u[l]) * mul
        if past_flow:
            ubfs[l] = up(bfs[l]) * mul
            sk_ub[l] = ubfs[l]
            for _ in range(2, LST):
                sk_ub[l] = up(sk_ub[l]) * mul
        if skip0:                                          # pwc.lua:423-429,462-471: the level's own maps are the outputs
            sk_u[l], sk_o[l] = fs[l], occs[l]
            if past_flow:
                sk_ub[l] = bfs[l]
        for f in (1, 3):
            if l > LST and f in frames:
                k = ff * (f - 2) if o["rescale_flow"] else ff * (f - 2) / 2 ** (l - 2)
                ws[f][l - 1] = warp_gather(cs[f][l - 1], ufs[l] * k).to(dtype)
            tmp = sk_ub[l] if (past_flow and f < 2) else sk_u[l]
            k2 = ff * (f - 2) if o["rescale_flow"] else ff * (f - 2) / 2 ** (l - LST)
            iws[f][l] = warp_gather(ds[f][l - LST + 1], tmp * k2).to(dtype)
    outs = []
    for l in range(LST, L + 1):
        outs.append(sk_u[l])
        if past_flow:
            outs.append(sk_ub[l])
        outs += [sk_o[l], iws[1][l], iws[3][l]]
    return [o_.numpy() for o_ in outs], {k: v.numpy() for k, v in inter.items()}
