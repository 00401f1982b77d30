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


def pwc_forward(x, wv, past_flow, dtype=torch.float64):
    """models/pwc.lua createModelMulti with the shipped opts, in torch ops.
    x: B x 9 x H x W numpy, wv: dict name->numpy (weights.views).  Returns the
    output table as a list of numpy arrays plus a dict of intermediates."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    x = t(x)
    P = {k: t(v) for k, v in wv.items()}
    Is = {f: x[:, 3 * (f - 1):3 * f] for f in (1, 2, 3)}
    ds = {}
    for f in (1, 3):
        ds[f] = {1: Is[f]}
        for k in range(2, 6):
            ds[f][k] = F.avg_pool2d(ds[f][k - 1], 2)
    cs = {}
    for f in (1, 2, 3):
        cs[f] = {1: Is[f]}
        for l in range(2, 8):
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
    fs, bfs, ufs, ubfs, sk_u, sk_ub, occs, sk_o, iws = {}, {}, {}, {}, {}, {}, {}, {}, {1: {}, 3: {}}
    inter = {}
    for l in range(7, 2, -1):
        src = cs if l == 7 else ws
        cvf = costvol_lua(cs[2][l], src[3][l], 9, True)
        cvb = costvol_lua(cs[2][l], src[1][l], 9, False)
        cv = torch.cat([cvf, cvb], 1)
        inter["cv%d" % l] = cv
        oin = [cv, cs[2][l]] + ([ufs[l + 1]] if l != 7 else [])
        occs[l] = F.softmax(dec(torch.cat(oin, 1), l, "occ"), dim=1)
        sk_o[l] = nn2(nn2(occs[l]))
        if l == 7:
            fs[l] = dec(cv, l, "flow")
            if past_flow:
                bfs[l] = dec(cv, l, "past")
        else:
            fs[l] = dec(torch.cat([cv, cs[2][l], ufs[l + 1]], 1), l, "flow")
            if past_flow:
                bfs[l] = dec(torch.cat([cv, cs[2][l], ubfs[l + 1]], 1), l, "past")
        inter["fs%d" % l] = fs[l]
        ufs[l] = up(fs[l]); sk_u[l] = up(ufs[l])
        if past_flow:
            ubfs[l] = up(bfs[l]); sk_ub[l] = up(ubfs[l])
        for f in (1, 3):
            if l > 3:
                ws[f][l - 1] = warp_gather(cs[f][l - 1], ufs[l] * (20.0 * (f - 2) / 2 ** (l - 2))).to(dtype)
            tmp = sk_ub[l] if (past_flow and f < 2) else sk_u[l]
            iws[f][l] = warp_gather(ds[f][l - 2], tmp * (20.0 * (f - 2) / 2 ** (l - 3))).to(dtype)
    outs = []
    for l in range(3, 8):
        outs.append(sk_u[l])
        if past_flow:
            outs.append(sk_ub[l])
        outs += [sk_o[l], iws[1][l], iws[3][l]]
    return [o.numpy() for o in outs], {k: v.numpy() for k, v in inter.items()}
