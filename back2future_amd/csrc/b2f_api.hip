// C-ABI layer of libb2f.so (include/b2f.h): context, weight packing, workspace, the forward
// pass of models/pwc.lua's graph (the host-buffer boundary around it: b2f_pipeline.hip), hipGraph
// capture, per-kernel HIP-event profiling, and op-level entry points for parity tests.
#include "b2f_ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace b2f;

static thread_local std::string g_err;
namespace b2f {
int api_fail(const std::string &m)
{
    g_err = m;
    return 1;
}
const std::string &api_error() { return g_err; }

// All activations are chunk-planar: [image][C/8][h][w][8].
ConvSeg cp8_seg(const float *ptr, int C, size_t hw)
{
    ConvSeg sgm;
    sgm.ptr = ptr;
    sgm.img_stride = (long)(hw * (size_t)((C + 7) / 8 * 8));
    sgm.chunk_stride = (long)(hw * 8);
    sgm.pix_stride = 8;
    sgm.nchunks = 0;
    return sgm;
}
}  // namespace b2f
static int fail(const std::string &m) { return api_fail(m); }

namespace {

int prof_id(b2f_ctx *c, const char *name)
{
    for (size_t i = 0; i < c->prof_names.size(); ++i)
        if (c->prof_names[i] == name) return (int)i;
    c->prof_names.push_back(name);
    c->prof_ms.push_back(0.0);
    c->prof_n.push_back(0);
    return (int)c->prof_names.size() - 1;
}

hipEvent_t get_event(b2f_ctx *c)
{
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// RAII bracket of one kernel launch with HIP events on the launch stream.
struct Scope {
    b2f_ctx *c;
    hipStream_t s;
    ProfEvent pe;
    bool on;
    Scope(b2f_ctx *c_, hipStream_t s_, const char *name, bool capturing) : c(c_), s(s_), on(c_->profile && !capturing)
    {
        if (on) {
            pe.name = prof_id(c, name);
            pe.a = get_event(c);
            pe.b = get_event(c);
            (void)hipEventRecord(pe.a, s);
        }
    }
    ~Scope()
    {
        if (on) {
            (void)hipEventRecord(pe.b, s);
            c->prof_pending.push_back(pe);
        }
    }
};

int prof_collect(b2f_ctx *c)
{
    for (ProfEvent &pe : c->prof_pending) {
        HIPCHK(hipEventSynchronize(pe.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
        c->prof_ms[pe.name] += ms;
        c->prof_n[pe.name] += 1;
        c->ev_pool.push_back(pe.a);
        c->ev_pool.push_back(pe.b);
    }
    c->prof_pending.clear();
    return 0;
}

// ---- weight packing ------------------------------------------------------------------
// Builds, for every conv of the canonical layout, the K-order the kernels consume:
//   features / inner decoder layers : input channels in order, zero padded to 8
//   decoder layer 1                 : segment 0 = cs[ref][l] (C_l channels),
//                                     segment 1 = cost-volume record (slot order in b2f_internal.h)
// while the Torch order of pwc.lua:308,334 is {cv 162, cs[ref][l] C_l, flow 2}.
int find_conv(const b2f_ctx *c, int kind, int level, int idx);

// B2F_WINO=0 falls back to the direct implicit-GEMM kernel everywhere (A/B runs, parity tests)
bool use_wino()
{
    static const bool on = !(getenv("B2F_WINO") && atoi(getenv("B2F_WINO")) == 0);
    return on;
}

// Kernel choice for a stride-1 layer: F(4x4) from B2F_WINO4_MIN_COUT outputs (default 32; 0 disables it),
// F(2x2) down to B2F_WINO_MIN_COUT (default 16), direct kernel below; 2 outputs: VALU kernel.
int wino_mode(int cout)
{
    static const int min4 = getenv("B2F_WINO4_MIN_COUT") ? atoi(getenv("B2F_WINO4_MIN_COUT")) : 32;
    if (use_wino() && cout == 2) return 1;           // last decoder layer: VALU kernel
    static const int min2 = getenv("B2F_WINO_MIN_COUT") ? atoi(getenv("B2F_WINO_MIN_COUT")) : 16;
    if (!use_wino() || cout < min2) return 0;
    return (min4 > 0 && cout >= min4 && cout % 4 == 0) ? 4 : 2;   // F(4x4) stores 4 channels at a time
}

int pack_all(b2f_ctx *c, const float *flat)
{
    const size_t n = c->lay.size();
    c->packed.assign(n, PackedConv());
    size_t total = 0;
    std::vector<std::vector<int>> maps(n);
    for (size_t i = 0; i < n; ++i) {
        const ConvDesc &d = c->lay[i];
        PackedConv &p = c->packed[i];
        p.cout = d.co;
        // stride-1 layers with >= 16 outputs run on the Winograd kernel (the first conv of a convUnit
        // has stride 2, the last decoder layer has 2 outputs: direct kernel)
        const bool stride1 = !(d.kind == KIND_FEAT && d.idx == 1 && d.level >= 2);   // (the level-1 unit of pwc_skip = 0 has stride 1, pwc.lua:172)
        p.wino = stride1 ? wino_mode(d.co) : 0;
        if (p.wino == 1 && d.kind != KIND_FEAT && d.idx == 1) p.wino = 0;   // two K segments: not for the narrow kernel
        if (stride1 && use_wino() && d.ci == 16 && d.co == 16) p.wino = 3;  // level-2 convUnit: dedicated kernel
        if (!stride1 && use_wino() && d.ci == 16 && d.co == 32) p.wino = 5;  // first conv of the level-3 convUnit: dedicated kernel
        if (p.wino == 4) { p.nt = 2; p.nblk = wino4_nblk(d.co); }
        else if (p.wino == 1 || p.wino == 3 || p.wino == 5) { p.nt = 1; p.nblk = 1; }
        else if (p.wino == 2) wino_choose_tiles(d.co, &p.nt, &p.nblk);
        else conv_choose_tiles(d.co, &p.nt, &p.nblk);
        std::vector<int> &m = maps[i];
        if (d.kind != KIND_FEAT && d.idx == 1 && c->g.shipped()) {
            const int Cl = kFeat[d.level];
            const bool has_feat = d.ci >= kND + Cl;          // level-7 flow decoder takes the cost volume only
            const bool has_flow = d.ci == kND + Cl + 2;
            if (has_feat) {
                p.nseg = 2;
                p.chunks[0] = Cl / kCK;
                for (int k = 0; k < Cl; ++k) m.push_back(kND + k);
            } else {
                p.nseg = 1;
            }
            p.chunks[p.nseg - 1] = kCvChunks;   // 21 chunks = 168 floats
            // record slots: 0..79 fwd 0..79 | 80..159 bwd 0..79 | 160 fwd80 | 161 bwd80 | 162,163 ufs | 164,165 ubfs
            const int flow_at = (d.kind == KIND_PAST) ? 164 : 162;   // pwc.lua:334 vs :337
            for (int k = 0; k < kCvRec; ++k) {
                int ci = -1;
                if (k < 80) ci = k;                       // fwd channel k      (Torch cv channel k)
                else if (k < 160) ci = 81 + (k - 80);     // bwd channel k - 80 (Torch cv channel 81 + ..)
                else if (k == 160) ci = 80;
                else if (k == 161) ci = 161;
                else if (has_flow && k >= flow_at && k < flow_at + 2) ci = kND + Cl + (k - flow_at);
                m.push_back(ci);
            }
        } else {
            p.nseg = 1;
            p.chunks[0] = (d.ci + kCK - 1) / kCK;
            for (int k = 0; k < p.chunks[0] * kCK; ++k) m.push_back(k < d.ci ? k : -1);
        }
        const int chunks = p.chunks[0] + (p.nseg > 1 ? p.chunks[1] : 0);
        total = (total + 3) & ~(size_t)3;   // every packing starts 16-byte aligned (conv_narrow2 reads its weights as float4)
        p.w_off = total;
        total += p.wino == 4 ? wino4_wpk_floats(chunks, p.nblk) : p.wino == 1 ? narrow2_wpk_floats(chunks)
                 : p.wino == 3 ? c16_wpk_floats() : p.wino == 5 ? c16s2_wpk_floats() : p.wino == 2 ? wino_wpk_floats(chunks, p.nt, p.nblk)
                 : conv_wpk_floats(chunks, p.nt, p.nblk);
        p.b_off = total;
        total += (size_t)p.nblk * p.nt * 32;
        if ((p.wino == 0 || (p.wino == 4 && c->bf16_conv >= 2)) && (d.co & 3) == 0) {   // direct layers: also packed for the kernel on the bf16 pipe (F(4x4)-class ones only while bf16_conv >= 2 reads that packing)
            total = (total + 3) & ~(size_t)3;
            p.w_off5 = total;
            total += convb_wpk_floats(chunks, d.co);
            p.b_off5 = total;
            total += (size_t)convb_nblk(d.co) * 64;
        }
        if (p.wino == 0 && !stride1 && (d.co & 31) == 0 && d.co <= 256) {   // stride-2 layers: the loader / consumer kernel's packing
            total = (total + 3) & ~(size_t)3;
            p.w_off7 = total;
            total += s2b_wpk_floats(chunks, d.co);
            p.b_off7 = total;
            total += (size_t)s2b_ntiles(d.co) * 32;
        }
        if (p.wino == 4 && c->wino6) {   // only while the option reads it (as the other optional packings)
            total = (total + 3) & ~(size_t)3;
            p.w_off8 = total;
            total += wino6_wpk_floats(chunks, d.co);
            p.b_off8 = total;
            total += (size_t)wino6_nblk(d.co) * 64;
        }
        if (p.wino == 4 && c->wino1d) {   // only while the option reads it (as the other optional packings)
            total = (total + 3) & ~(size_t)3;
            p.w_off6 = total;
            total += w1b_wpk_floats(chunks, d.co);
            p.b_off6 = total;
            total += (size_t)w1b_nblk(d.co) * 64;
        }
        if (p.wino == 4) {
            wino_choose_tiles(d.co, &p.nt2, &p.nblk2);
            p.w_off2 = total;
            total += wino_wpk_floats(chunks, p.nt2, p.nblk2);
            p.b_off2 = total;
            total += (size_t)p.nblk2 * p.nt2 * 32;
#if B2F_EXPERIMENTS
            if (c->wino2_split) {
                total = (total + 3) & ~(size_t)3;
                p.w_off4 = total;
                total += wino2s_wpk_floats(chunks, p.nblk);
            }
            if (c->wino4_split || c->wino4_hybrid) { // only while an option that reads it is on: 1.5x the F(4x4) packing
                total = (total + 3) & ~(size_t)3;    // 16-byte aligned: the split weights are read with dwordx4 loads
                p.w_off3 = total;
                total += wino4s_wpk_floats(chunks, p.nblk);
            }
#endif
        }
    }
    c->first_w_off = total; total += 27 * 16;
    c->first_b_off = total; total += 16;
    std::vector<float> host(total, 0.f);
    const int id_first = find_conv(c, KIND_FEAT, 2, 1);   // Torch 16 x 3 x 3 x 3 -> [tap (c,ky,kx)][cout]
    if (id_first >= 0 && c->lay[(size_t)id_first].ci == 3 && c->lay[(size_t)id_first].co == 16) {   // (absent with pwc_siamese = 0, 16 -> 16 with pwc_skip = 0)
        const ConvDesc &d = c->lay[(size_t)id_first];
        for (int o = 0; o < 16; ++o) {
            for (int t = 0; t < 27; ++t) host[c->first_w_off + (size_t)t * 16 + o] = flat[d.w_off + (size_t)o * 27 + t];
            host[c->first_b_off + o] = flat[d.b_off + o];
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const ConvDesc &d = c->lay[i];
        const PackedConv &p = c->packed[i];
        const int chunks = p.chunks[0] + (p.nseg > 1 ? p.chunks[1] : 0);
        if (p.wino == 4) {
            wino4_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, p.nblk,
                               host.data() + p.w_off, host.data() + p.b_off);
            wino_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, p.nt2, p.nblk2,
                              host.data() + p.w_off2, host.data() + p.b_off2);
            if (p.w_off5) convb_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, host.data() + p.w_off5, host.data() + p.b_off5);
#if B2F_EXPERIMENTS
            if (p.w_off4) wino2s_pack_weights(flat + d.w_off, d.co, d.ci, maps[i].data(), chunks, p.nblk, host.data() + p.w_off4);
            if (p.w_off3) wino4s_pack_weights(flat + d.w_off, d.co, d.ci, maps[i].data(), chunks, p.nblk, host.data() + p.w_off3);
#endif
            if (p.w_off6) w1b_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, host.data() + p.w_off6, host.data() + p.b_off6);
            if (p.w_off8) wino6_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, host.data() + p.w_off8, host.data() + p.b_off8);
        } else if (p.wino == 1)
            narrow2_pack_weights(flat + d.w_off, flat + d.b_off, d.ci, maps[i].data(), chunks, host.data() + p.w_off,
                                 host.data() + p.b_off);
        else if (p.wino == 3)
            c16_pack_weights(flat + d.w_off, flat + d.b_off, d.ci, maps[i].data(), host.data() + p.w_off, host.data() + p.b_off);
        else if (p.wino == 5)
            c16s2_pack_weights(flat + d.w_off, flat + d.b_off, d.ci, maps[i].data(), host.data() + p.w_off, host.data() + p.b_off);
        else if (p.wino == 2)
            wino_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, p.nt, p.nblk,
                              host.data() + p.w_off, host.data() + p.b_off);
        else {
            conv_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, p.nt, p.nblk,
                              host.data() + p.w_off, host.data() + p.b_off);
            if (p.w_off5) convb_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, host.data() + p.w_off5, host.data() + p.b_off5);
            if (p.w_off7) s2b_pack_weights(flat + d.w_off, flat + d.b_off, d.co, d.ci, maps[i].data(), chunks, host.data() + p.w_off7, host.data() + p.b_off7);
        }
    }
    if (c->wpk_floats != total) {
        if (c->wpk_dev) HIPCHK(hipFree(c->wpk_dev));
        c->wpk_dev = nullptr;
        HIPCHK(hipMalloc(&c->wpk_dev, total * sizeof(float)));
        c->wpk_floats = total;
    }
    HIPCHK(hipMemcpy(c->wpk_dev, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int find_conv(const b2f_ctx *c, int kind, int level, int idx)
{
    for (size_t i = 0; i < c->lay.size(); ++i)
        if (c->lay[i].kind == kind && c->lay[i].level == level && c->lay[i].idx == idx) return (int)i;
    return -1;
}

// ---- workspace -------------------------------------------------------------------------
struct Plan {
    int B, H, W;
    bool full;          // full model:forward table (all decoders, image pyramid, image warps)
    int rec;            // cost-volume record size in floats
    int h[8], w[8];
    size_t img, tmp, cs[8], U[8], UB[8], cv, d[6], fs, bfs, logits, u2, flow_planar, ds[6], total;
};

Plan make_plan(int B, int H, int W, bool full, bool past_flow)
{
    Plan p;
    p.B = B; p.H = H; p.W = W;
    p.full = full;
    p.rec = kCvRec;
    for (int l = 1; l <= 7; ++l) { p.h[l] = H >> (l - 1); p.w[l] = W >> (l - 1); }
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~(size_t)63; return o; };
    p.img = full ? take((size_t)3 * B * H * W * kImgC) : 0;   // packed frames: only the full table needs them
    p.tmp = take((size_t)3 * B * p.h[2] * p.w[2] * kFeat[2]);
    for (int l = 2; l <= 7; ++l) p.cs[l] = take((size_t)3 * B * p.h[l] * p.w[l] * kFeat[l]);
    for (int l = 3; l <= 6; ++l) p.U[l] = take((size_t)B * p.h[l] * p.w[l] * 2);
    for (int l = 3; l <= 6; ++l) p.UB[l] = (full && past_flow) ? take((size_t)B * p.h[l] * p.w[l] * 2) : 0;
    p.cv = take((size_t)B * p.h[3] * p.w[3] * p.rec + 64);
    const size_t px3 = (size_t)B * p.h[3] * p.w[3];
    for (int i = 1; i <= 5; ++i) p.d[i] = take(px3 * kDec[i]);
    p.fs = take(px3 * 8);       // conv outputs are chunk-planar: 2 channels live in one 8-float chunk
    p.bfs = take(px3 * 8);
    p.logits = take(px3 * 8);
    p.u2 = take(px3 * 4 * 2);
    p.flow_planar = take((size_t)B * 2 * H * W);
    // image pyramid ds[f][k], f in {1,3}, k = 2..5 (pwc.lua:148-158): [2][B][H/2^(k-1)][W/2^(k-1)][8]
    for (int k = 2; k <= 5; ++k) p.ds[k] = full ? take((size_t)2 * B * (H >> (k - 1)) * (W >> (k - 1)) * kImgC) : 0;
    p.total = off;
    return p;
}

}  // namespace
void b2f::drop_gen_out(b2f_ctx *c)
{
    for (float *p : c->gen_out)
        if (p) (void)hipFree(p);
    c->gen_out.clear();
    c->genB = c->genH = c->genW = 0;
}
void b2f::drop_graphs(b2f_ctx *c)
{
    for (auto &g : c->graphs)
        if (g.second) (void)hipGraphExecDestroy(g.second);
    c->graphs.clear();
}
namespace {

int ensure_workspace_floats(b2f_ctx *c, size_t total)
{
    if (total > c->arena_floats) {
        // rare (a larger shape than any before): full device syncs on both sides.  Earlier passes may have run on a
        // caller's stream, and hipMemset is asynchronous on the null stream, which the non-blocking streams the
        // kernels run on do not wait for -- without the second sync the memset could land on top of the first
        // kernels' output (found by tools/soak.py).
        if (c->arena) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(c->arena));
            c->arena = nullptr;
            c->arena_floats = 0;
            drop_graphs(c);
        }
        HIPCHK(hipMalloc(&c->arena, total * sizeof(float)));
        HIPCHK(hipMemset(c->arena, 0, total * sizeof(float)));
        HIPCHK(hipDeviceSynchronize());
        c->arena_floats = total;
    }
    return 0;
}
int ensure_workspace(b2f_ctx *c, const Plan &p) { return ensure_workspace_floats(c, p.total); }

// ---- one conv launch from the packed table -----------------------------------------------
// All activations are chunk-planar: [image][C/8][h][w][8].
int run_conv(b2f_ctx *c, hipStream_t s, bool cap, int conv_id, const ConvSeg *segs, int nimg, int H, int W,
             int stride, int leaky, float *out)
{
    const PackedConv &p = c->packed[conv_id];
    // kernel for this call: F(4x4) layers fall back to their F(2x2) packing when the launch would leave most of
    // the chip idle.  An F(4x4) block (16 x 32 pixels, one per CU) takes about three times as long as an F(2x2)
    // block (8 x 16 pixels) that has its CU to itself and 1.5 times as long as one that shares it with a second
    // block, so compare the number of block rounds each kernel needs on 256 CUs.
    // That rule makes a triplet's result depend (at 1e-6 level) on the batch it is computed in, so it is opt-in
    // (option adaptive_kernels); the default rule looks at the map size only: F(2x2) below wino4_min_pixels.
    bool alt = false, split = false;
    if (p.wino == 4) {
        // adaptive_kernels: -1 (default) = per launch for single-triplet calls (the reference's own calling pattern, back2future.lua:73:
        // latency matters and most launches leave the chip half empty), by map size for batches; 0 never, 1 always per launch
        const bool per_launch = c->adaptive_kernels > 0 || (c->adaptive_kernels < 0 && c->cur_batch == 1);
        if (!per_launch) {
            alt = H * W < c->wino4_min_pixels;
            split = alt && p.nt2 == 2 && H * W <= c->wino_split_pixels;
        } else {
            // cost in twentieths of a lone F(2x2) block: F(4x4) block 60; F(2x2) block 20 alone, 40 per pair sharing a CU;
            // F(2x2) block that computes one of the two N tiles only 13 / 26
            const long tiles2 = (long)nimg * ((H + 7) / 8) * ((W + 15) / 16);
            const long b4 = (long)nimg * ((H + 15) / 16) * ((W + 31) / 32) * p.nblk;
            const long b2 = tiles2 * p.nblk2, b2s = tiles2 * ((p.cout + 31) / 32);
            const long t4 = 60 * ((b4 + 255) / 256);
            const long t2 = b2 <= 256 ? 20 : 40 * ((b2 + 511) / 512);
            const long t2s = p.nt2 != 2 ? t2 : b2s <= 256 ? 13 : 26 * ((b2s + 511) / 512);
            alt = std::min(t2, t2s) < t4;
            split = alt && t2s < t2;
        }
    }
    const int mode = alt ? 2 : p.wino;
    const int nt = alt ? p.nt2 : p.nt, nblk = alt ? p.nblk2 : p.nblk;
    ConvLaunch L;
    L.nseg = p.nseg;
    for (int i = 0; i < p.nseg; ++i) {
        L.seg[i] = segs[i];
        L.seg[i].nchunks = p.chunks[i];
    }
    if (p.nseg == 1) L.seg[1] = L.seg[0], L.seg[1].nchunks = 0;
    L.wpk = c->wpk_dev + (alt ? p.w_off2 : p.w_off);
    L.bias = c->wpk_dev + (alt ? p.b_off2 : p.b_off);
    L.out = out;
    L.cout = p.cout;
    L.nt = nt;
    L.nblk = nblk;
    L.H = H; L.W = W; L.stride = stride;
    L.Ho = (H + 2 - 3) / stride + 1;
    L.Wo = (W + 2 - 3) / stride + 1;
    L.out_img_stride = (long)((size_t)L.Ho * L.Wo * ((p.cout + 7) / 8 * 8));
    L.out_chunk_stride = (long)((size_t)L.Ho * L.Wo * 8);
    L.out_pix_stride = 8;
    L.nimg = nimg;
    L.nsplit = split ? 1 : 0;
    L.leaky = leaky;
    L.tiles_per_block = c->s2_tiles_per_block;
    L.w4_persist = c->wino4_persistent;
    L.w8 = c->wino8;
#if B2F_EXPERIMENTS
    L.wpk_split = (mode == 4 && (c->wino4_split || c->wino4_hybrid) && p.w_off3) ? c->wpk_dev + p.w_off3 : nullptr;
    L.w4_hybrid = c->wino4_hybrid;
    L.wpk_split2 = (mode == 4 && c->wino2_split && p.w_off4) ? c->wpk_dev + p.w_off4 : nullptr;
#endif
    L.bf16_direct = c->bf16_direct;
    bool bf6 = p.w_off5 && ((mode == 0 && c->bf16_conv) || (mode == 4 && stride == 1 && H * W >= c->bf16_conv_min_pixels && ((c->bf16_conv == 2 && p.cout <= 32) || c->bf16_conv >= 3)));
    if (bf6) {
        L.wpk_bf6 = c->wpk_dev + p.w_off5; L.bias_bf6 = c->wpk_dev + p.b_off5;
        bf6 = convb_supported(L);            // the profile row below names the kernel that really runs
        if (!bf6) { L.wpk_bf6 = nullptr; L.bias_bf6 = nullptr; }
    }
    bool s2l = false;
    // (layers of fewer than 64 input channels keep conv3x3_bf6: with four short chunks per tile the loader / consumer block is bound by
    // its weight traffic -- 2 KB per six MFMAs -- 32 -> 64 measured 0.45 against 0.43 ms; s2_loader = 2 sends them there too)
    if (mode == 0 && stride == 2 && c->bf16_conv && c->s2_loader && p.w_off7 && (c->s2_loader >= 2 || p.chunks[0] + (p.nseg > 1 ? p.chunks[1] : 0) >= 8)) {
        L.wpk_s2b = c->wpk_dev + p.w_off7; L.bias_s2b = c->wpk_dev + p.b_off7;
        s2l = s2b_supported(L);
        if (s2l) { bf6 = false; L.nsplit = c->s2_tile_groups ? 0 : -1; }
    }
    // wino1d = 1: the n-blocks with more than 32 real outputs on the 1-D Winograd bf16 kernel, a last block of <= 32 outputs on the
    // F(4x4) single-N-tile kernel (half the bf16 kernel's MFMAs would multiply zero padding); 2: every n-block
    const int w1d_blocks = c->wino1d >= 2 ? w1b_nblk(p.cout) : p.cout / 64 + (p.cout % 64 > 32 ? 1 : 0);
    bool w1d = !bf6 && mode == 4 && stride == 1 && c->wino1d && p.w_off6 && w1d_blocks > 0;
    if (w1d) {
        L.wpk_w1b = c->wpk_dev + p.w_off6; L.bias_w1b = c->wpk_dev + p.b_off6;
        L.w1b_nblk = w1d_blocks;
        w1d = w1b_supported(L);
    }
    // wino6 = 1: Winograd F(6x6) (blocks of 64 outputs, a last block of 32 when the outputs are <= 32 mod 64)
    bool w6 = !bf6 && !w1d && mode == 4 && stride == 1 && c->wino6 && p.w_off8 && H * W >= c->wino6_min_pixels;
    if (w6) {
        L.wpk_w6 = c->wpk_dev + p.w_off8; L.bias_w6 = c->wpk_dev + p.b_off8;
        w6 = wino6_supported(L);
    }
    char name[48];
    const bool per_layer = c->profile_layers != 0;   // one profile row per (layer shape, map size)
    if (per_layer)
        snprintf(name, sizeof name, "conv%s_%dto%d_%dx%d", s2l ? "L2" : bf6 ? (stride == 1 ? "E1" : "E2") : w1d ? "V1" : w6 ? "W6" : mode == 4 ? "W4" : mode == 3 ? ((B2F_EXPERIMENTS && c->bf16_direct) ? "B16" : "C16") : mode == 5 ? "S16" : mode == 2 ? "W2" : mode == 1 ? "N2" : stride == 1 ? "D1" : "D2",
                 (p.chunks[0] + (p.nseg > 1 ? p.chunks[1] : 0)) * 8, p.cout, H, W);
    else
        snprintf(name, sizeof name, s2l ? "conv3x3_s2b_%d" : bf6 ? (stride == 1 ? "conv3x3_s1_bf16_%d" : "conv3x3_s2_bf16_%d") : w1d ? "conv3x3_w1b_%d" : w6 ? "conv3x3_wino6_nt%d" : mode == 4 ? "conv3x3_wino4_nt%d" : mode == 3 ? "conv3x3_c16_%d" : mode == 5 ? "conv3x3_s2x16_%d" : mode == 2 ? "conv3x3_wino_nt%d"
                                    : mode == 1 ? "conv3x3_narrow%d" : (stride == 1 ? "conv3x3_s1_nt%d" : "conv3x3_s2_nt%d"),
                 mode == 1 ? 2 : nt);
    Scope sc(c, s, name, cap);
    if (s2l) HIPCHK(launch_conv3x3_s2b(L, s));
    else if (bf6) HIPCHK(launch_conv3x3_bf6(L, s));
    else if (w1d) {
        HIPCHK(launch_conv3x3_w1b(L, s));
        if (w1d_blocks < w1b_nblk(p.cout)) HIPCHK(launch_conv3x3_wino4_rem(L, s));
    }
    else if (w6) HIPCHK(launch_conv3x3_wino6(L, s));
    else if (mode == 4) HIPCHK(launch_conv3x3_wino4(L, s));
    else if (mode == 1) HIPCHK(launch_conv_narrow2(L, s));
    else if (mode == 3) HIPCHK(launch_conv3x3_c16(L, s));
    else if (mode == 5) HIPCHK(launch_conv3x3_c16s2(L, s));
    else if (mode == 2) HIPCHK(launch_conv3x3_wino(L, s));
    else HIPCHK(launch_conv3x3(L, s));
    return 0;
}

// decoder(n) of pwc.lua:76-85 at level l; input = {cs[ref][l], cost-volume record}
int run_decoder(b2f_ctx *c, hipStream_t s, bool cap, const Plan &P, int kind, int l, float *out2)
{
    const size_t *dbuf = P.d;
    float *A = c->arena;
    const int h = P.h[l], w = P.w[l], B = P.B;
    const size_t hw = (size_t)h * w;
    const int Cl = kFeat[l];
    const int id1 = find_conv(c, kind, l, 1);
    if (id1 < 0) return fail("decoder not present in this model");
    ConvSeg segs[2];
    const ConvSeg seg_ref = cp8_seg(A + P.cs[l] + (size_t)1 * B * hw * Cl, Cl, hw);
    const ConvSeg seg_cv = cp8_seg(A + P.cv, kCvRec, hw);
    if (c->packed[id1].nseg == 2) { segs[0] = seg_ref; segs[1] = seg_cv; }
    else { segs[0] = seg_cv; segs[1] = seg_cv; }
    CHK(run_conv(c, s, cap, id1, segs, B, h, w, 1, 1, A + dbuf[1]));
    for (int i = 2; i <= 6; ++i) {
        const ConvSeg in = cp8_seg(A + dbuf[i - 1], kDec[i - 1], hw);
        float *o = (i == 6) ? out2 : A + dbuf[i];
        CHK(run_conv(c, s, cap, find_conv(c, kind, l, i), &in, B, h, w, 1, i < 6, o));
    }
    return 0;
}

// Device outputs of one forward.  Pruned mode (computeFlow): flow / occ / est3 of the finest
// level.  Full mode: the whole output table of pwc.lua:459-489, per level l = 3..7 (index l):
// skip_ufs, skip_ubfs (Soft), skip_occs, iws[1], iws[3], all planar at 4h_l x 4w_l.
struct Outs {
    float *flow = nullptr, *occ = nullptr, *est3 = nullptr;
    float *t_ufs[8] = {nullptr}, *t_ubfs[8] = {nullptr}, *t_occ[8] = {nullptr}, *t_iw1[8] = {nullptr}, *t_iw3[8] = {nullptr};
};

// The computeFlow graph: pruned to the live set or, with P.full, the
// complete model:forward of models/pwc.lua.
int forward_impl(b2f_ctx *c, hipStream_t s, bool cap, const void *dev_in, int in_kind, const Plan &P, const Outs &O)
{
    c->cur_batch = c->req_batch > 0 ? c->req_batch : P.B;
    float *A = c->arena;
    const int B = P.B;
    const bool full = P.full, past = c->past_flow && full;
    const int unit = in_kind == B2F_IN_UNIT;
    if (full) {   // the packed frames are only needed for the image pyramid / image warps of the full table
        Scope sc(c, s, "pack_input", cap);
        HIPCHK(launch_pack_input((const float *)dev_in, unit, B, P.H, P.W, A + P.img, s));
    }
    // siamese feature pyramid, the three frames batched (shared weights, pwc.lua:169-211)
    // (the head kernel reads the c16 / c16s2 packings of its two layers: B2F_WINO=0 packs them for the direct kernel instead)
    const int head_id1 = find_conv(c, KIND_FEAT, 2, 2), head_id2 = find_conv(c, KIND_FEAT, 3, 1);
    const bool head_fused = c->bf16_direct >= 2 && P.h[2] >= 4 && P.w[2] >= 4 && head_id1 >= 0 && head_id2 >= 0 &&
                            c->packed[(size_t)head_id1].wino == 3 && c->packed[(size_t)head_id2].wino == 5;   // level-2 conv 2 + level-3 conv 1 as one streaming kernel; cs[2] then holds the 32-channel level-3 map
    for (int l = 2; l <= 7; ++l) {
        const int hi = P.h[l - 1], wi = P.w[l - 1], ho = P.h[l], wo = P.w[l];
        const int Ci = (l == 2) ? kImgC : kFeat[l - 1], Co = kFeat[l];
        if (l == 2) {
            Scope sc(c, s, "conv_first", cap);
            HIPCHK(launch_conv_first((const float *)dev_in, unit, B, P.H, P.W, c->wpk_dev + c->first_w_off,
                                     c->wpk_dev + c->first_b_off, A + P.tmp, s));
        } else if (l == 3 && head_fused) {   // conv 1 of level 3 ran inside the fused head: its output sits in the (otherwise unused) cs[2] region
            const ConvSeg in2f = cp8_seg(A + P.cs[2], Co, (size_t)ho * wo);
            CHK(run_conv(c, s, cap, find_conv(c, KIND_FEAT, l, 2), &in2f, 3 * B, ho, wo, 1, 1, A + P.cs[l]));
            continue;
        } else {
            const ConvSeg in1 = cp8_seg(A + P.cs[l - 1], Ci, (size_t)hi * wi);
            CHK(run_conv(c, s, cap, find_conv(c, KIND_FEAT, l, 1), &in1, 3 * B, hi, wi, 2, 1, A + P.tmp));
        }
        if (l == 2 && head_fused) {   // level-2 conv 2 + level-3 conv 1 in one streaming kernel (b2f_head.hip)
            const PackedConv &p1 = c->packed[find_conv(c, KIND_FEAT, 2, 2)], &p2 = c->packed[find_conv(c, KIND_FEAT, 3, 1)];
            HeadLaunch hl;
            hl.in = A + P.tmp; hl.in_img_stride = (long)((size_t)ho * wo * 16); hl.in_chunk_stride = (long)((size_t)ho * wo * 8); hl.in_pix_stride = 8;
            hl.H1 = ho; hl.W1 = wo;
            hl.w1 = c->wpk_dev + p1.w_off; hl.b1 = c->wpk_dev + p1.b_off; hl.w2 = c->wpk_dev + p2.w_off; hl.b2 = c->wpk_dev + p2.b_off;
            hl.out = A + P.cs[2];
            hl.Ho = P.h[3]; hl.Wo = P.w[3]; hl.nimg = 3 * B;
            hl.out_img_stride = (long)((size_t)hl.Ho * hl.Wo * 32); hl.out_chunk_stride = (long)((size_t)hl.Ho * hl.Wo * 8); hl.out_pix_stride = 8;
            char hname[48];
            snprintf(hname, sizeof hname, c->profile_layers ? "convH16_16to32_%dx%d" : "conv_head16_bf16", ho, wo);
            Scope sc(c, s, hname, cap);
            HIPCHK(launch_conv_head16(hl, s));
            continue;
        }
        const ConvSeg in2 = cp8_seg(A + P.tmp, Co, (size_t)ho * wo);
        CHK(run_conv(c, s, cap, find_conv(c, KIND_FEAT, l, 2), &in2, 3 * B, ho, wo, 1, 1, A + P.cs[l]));
    }
    if (full) {
        // image pyramid of frames 1 and 3 for the warped-image outputs (pwc.lua:148-158);
        // ds[k] holds [frame 1 | frame 3], level 1 is the packed input itself
        Scope sc(c, s, "avgpool2", cap);
        const size_t img_f = (size_t)B * P.H * P.W * kImgC;
        for (int f = 0; f < 2; ++f) {
            const float *src = A + P.img + (size_t)(f == 0 ? 0 : 2) * img_f;
            for (int k = 2; k <= 5; ++k) {
                const int hk = P.H >> (k - 2), wk = P.W >> (k - 2);
                float *dst = A + P.ds[k] + (size_t)f * B * (hk / 2) * (wk / 2) * kImgC;
                HIPCHK(launch_avgpool2_nhwc(src, B, hk, wk, kImgC, dst, s));
                src = dst;
            }
        }
    }
    for (int l = 7; l >= 3; --l) {   // pwc.lua:237
        const int h = P.h[l], w = P.w[l], Cl = kFeat[l];
        const size_t hw = (size_t)h * w;
        CorrLaunch cl;
        cl.ref = A + P.cs[l] + (size_t)1 * B * hw * Cl;
        cl.nbr_fut = A + P.cs[l] + (size_t)2 * B * hw * Cl;
        cl.nbr_past = A + P.cs[l];
        cl.img_stride = (long)(hw * Cl);
        cl.chunk_stride = (long)(hw * 8);
        cl.pix_stride = 8;
        cl.flow = (l < 7) ? A + P.U[l] : nullptr;
        cl.flow_b = (l < 7 && past) ? A + P.UB[l] : nullptr;
        cl.k = (float)(20.0 / std::pow(2.0, l - 1));   // nn.MulConstant(20*(f-ref)/2^(l-2)) one level up, pwc.lua:404
        cl.out = A + P.cv;
        cl.out_img_stride = (long)(hw * kCvRec);
        cl.out_chunk_stride = (long)(hw * 8);
        cl.out_pix_stride = 8;
        cl.B = B; cl.C = Cl; cl.h = h; cl.w = w;
        cl.variant = c->corr_variant;
        cl.ablate = c->corr_ablate;
        {
            char cname[48];
            snprintf(cname, sizeof cname, c->profile_layers ? "warp_costvol_%dx%d" : "warp_costvol", h, w);
            Scope sc(c, s, cname, cap);
            HIPCHK(launch_warp_costvol(cl, s));
        }
        // occlusion decoder + SpatialSoftMax + nearest x4 (pwc.lua:288-321); dead below level 3 unless full
        float *occ_out = full ? O.t_occ[l] : (l == 3 ? O.occ : nullptr);
        float *occ_out2 = (!full && l == 3 && c->past_flow) ? O.est3 : nullptr;   // Soft: est[3] = skip_occs[3]
        if (occ_out || occ_out2) {
            // (round 3 ran this decoder on a side stream beside the flow decoder: both chains are chip-filling persistent
            // kernels, measured no gain, and the second set of intermediates cost a third of the arena -- removed in round 4)
            CHK(run_decoder(c, s, cap, P, KIND_OCC, l, A + P.logits));
            {
                Scope sc(c, s, "softmax_nearest4", cap);
                if (occ_out) HIPCHK(launch_softmax_nearest4_planar(A + P.logits, 8, B, h, w, occ_out, s));
                if (occ_out2) HIPCHK(launch_softmax_nearest4_planar(A + P.logits, 8, B, h, w, occ_out2, s));
            }
        }
        CHK(run_decoder(c, s, cap, P, KIND_FLOW, l, A + P.fs));
        if (past) CHK(run_decoder(c, s, cap, P, KIND_PAST, l, A + P.bfs));
        // upsampling (pwc.lua:359-390): ufs = bilinear x2; skip_ufs = a second bilinear x2
        float *skip_f = full ? O.t_ufs[l] : (l == 3 ? (O.flow ? O.flow : A + P.flow_planar) : nullptr);
        {
            Scope sc(c, s, "upsample_flow2x", cap);
            float *u = (l > 3) ? A + P.U[l - 1] : A + P.u2;
            HIPCHK(launch_upsample_flow2x(A + P.fs, 8, B, h, w, u, s));
            if (skip_f) HIPCHK(launch_upsample_flow2x_planar(u, 2, B, 2 * h, 2 * w, skip_f, s));
            if (past) {
                // the past flow of level 3 has no next level: its x2 goes through the second half of u2's slot
                float *ub = (l > 3) ? A + P.UB[l - 1] : A + P.d[1];
                HIPCHK(launch_upsample_flow2x(A + P.bfs, 8, B, h, w, ub, s));
                if (O.t_ubfs[l]) HIPCHK(launch_upsample_flow2x_planar(ub, 2, B, 2 * h, 2 * w, O.t_ubfs[l], s));
            }
        }
        if (full) {
            // warped images iws[f][l] = warp(ds[f][l-2], skip_u(b)fs[l] * 20(f-2)/2^(l-3))  (pwc.lua:422-446)
            Scope sc(c, s, "warp_image", cap);
            const int k = l - 2;   // pyramid level of the image (1 = full resolution)
            const int hk = 4 * h, wk = 4 * w;
            const size_t img_f = (size_t)B * hk * wk * kImgC;
            const float *im1 = (k == 1) ? A + P.img : A + P.ds[k];
            const float *im3 = (k == 1) ? A + P.img + 2 * img_f : A + P.ds[k] + img_f;
            const float kk = (float)(20.0 / std::pow(2.0, l - 3));
            const float *fl1 = (past && O.t_ubfs[l]) ? O.t_ubfs[l] : O.t_ufs[l];
            if (O.t_iw1[l]) HIPCHK(launch_warp_image_planar(im1, fl1, -kk, B, hk, wk, O.t_iw1[l], s));
            if (O.t_iw3[l]) HIPCHK(launch_warp_image_planar(im3, O.t_ufs[l], kk, B, hk, wk, O.t_iw3[l], s));
        }
    }
    if (!full && O.est3 && !c->past_flow) {
        // Hard: est[3] = iws[1][3] = warp(I1, skip_ufs[3] * 20*(1-2)/2^0)  (pwc.lua:422-446,459-489)
        Scope sc(c, s, "warp_image", cap);
        HIPCHK(launch_warp_input_planar((const float *)dev_in, unit, 0, O.flow ? O.flow : A + P.flow_planar, -20.0f, B, P.H, P.W,
                                        O.est3, s));
    }
    return 0;
}

}  // namespace
int b2f::find_conv_id(const b2f_ctx *c, int kind, int level, int idx) { return find_conv(c, kind, level, idx); }
int b2f::run_conv_layer(b2f_ctx *c, hipStream_t s, bool cap, int conv_id, const ConvSeg *segs, int nimg, int H, int W, int stride,
                        int leaky, float *out)
{
    if (conv_id < 0) return fail("b2f: layer not present in this model");
    return run_conv(c, s, cap, conv_id, segs, nimg, H, W, stride, leaky, out);
}
int b2f::ensure_arena(b2f_ctx *c, size_t floats) { return ensure_workspace_floats(c, floats); }
int b2f::check_shape(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return fail("b2f: non-positive shape");
    if (H % 64 || W % 64) return fail("b2f: H and W must be multiples of 64 (7 pyramid levels, back2future.lua:54-67)");
    return 0;
}
namespace {

bool parse_random(const char *name, bool *past, unsigned long long *seed, float *gain)
{
    if (strncmp(name, "random:", 7) != 0) return false;
    std::string s(name + 7);
    std::vector<std::string> parts;
    size_t pos = 0;
    while (true) {
        size_t q = s.find(':', pos);
        parts.push_back(s.substr(pos, q == std::string::npos ? q : q - pos));
        if (q == std::string::npos) break;
        pos = q + 1;
    }
    if (parts[0] == "hard") *past = false;
    else if (parts[0] == "soft") *past = true;
    else return false;
    *seed = 2;   // manualSeed default, opts.lua:27
    *gain = 1.f;
    if (parts.size() > 1 && !parts[1].empty()) *seed = strtoull(parts[1].c_str(), nullptr, 10);
    if (parts.size() > 2 && !parts[2].empty()) *gain = strtof(parts[2].c_str(), nullptr);
    return true;
}

bool ends_with(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int install_weights(b2f_ctx *c, const float *flat, long long n, bool past)
{
    GraphOpts g = c->g;
    g.past_flow = past;
    if (n != param_count(g)) return fail("b2f: weight count does not match the architecture");
    // captured graphs hold the packed-weight pointers and the Hard / Soft topology of the moment they were captured
    HIPCHK(hipDeviceSynchronize());
    drop_graphs(c);
    drop_gen_out(c);
    if (c->w_dev && (c->nparams != n)) { HIPCHK(hipFree(c->w_dev)); c->w_dev = nullptr; }
    c->past_flow = past;
    c->g = g;
    c->nparams = n;
    long long t = 0;
    c->lay = weight_layout(g, &t);
    if (!c->w_dev) HIPCHK(hipMalloc(&c->w_dev, (size_t)n * sizeof(float)));
    HIPCHK(hipMemcpy(c->w_dev, flat, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return pack_all(c, flat);
}

}  // namespace

// ======================================================================================
extern "C" {

const char *b2f_last_error(void) { return g_err.c_str(); }
int b2f_version(void) { return 1000; }

long long b2f_param_count(int past_flow) { return param_count(past_flow != 0); }

int b2f_random_weights(unsigned long long seed, int past_flow, float gain, float *out, long long n) try
{
    if (!out || n != param_count(past_flow != 0)) return fail("b2f_random_weights: bad buffer size");
    random_weights(seed, past_flow != 0, gain, out);
    return 0;
}
B2F_CATCH("b2f_random_weights")

int b2f_load_t7(const char *path, float *out, long long cap, long long *n, int *past_flow) try
{
    std::vector<float> flat;
    bool past = false;
    std::string err;
    if (!path) return fail("b2f_load_t7: null path");
    if (!load_t7(path, flat, past, err)) return fail("b2f_load_t7: " + err);
    if (n) *n = (long long)flat.size();
    if (past_flow) *past_flow = past ? 1 : 0;
    if (out) {
        if (cap < (long long)flat.size()) return fail("b2f_load_t7: output buffer too small");
        memcpy(out, flat.data(), flat.size() * sizeof(float));
    }
    return 0;
}
B2F_CATCH("b2f_load_t7")

int b2f_load_t7_ex(const char *path, const char *graph_opts, float *out, long long cap, long long *n, char *opts_out, int opts_cap) try
{
    if (!path) return fail("b2f_load_t7_ex: null path");
    GraphOpts g;
    std::string err;
    if (!parse_graph_opts(graph_opts, g, err)) return fail("b2f_load_t7_ex: " + err);
    std::vector<float> flat;
    if (!load_t7_ex(path, g, !(graph_opts && *graph_opts), flat, err)) return fail("b2f_load_t7_ex: " + err);
    if (n) *n = (long long)flat.size();
    if (opts_out && opts_cap > 0) {
        const std::string t = graph_opts_string(g);
        if ((int)t.size() + 1 > opts_cap) return fail("b2f_load_t7_ex: option buffer too small");
        memcpy(opts_out, t.c_str(), t.size() + 1);
    }
    if (out) {
        if (cap < (long long)flat.size()) return fail("b2f_load_t7_ex: output buffer too small");
        memcpy(out, flat.data(), flat.size() * sizeof(float));
    }
    return 0;
}
B2F_CATCH("b2f_load_t7_ex")

int b2f_init(const char *name_or_path, int device, b2f_ctx **out) { return b2f_init_ex(name_or_path, device, nullptr, out); }

int b2f_init_ex(const char *name_or_path, int device, const char *graph_opts, b2f_ctx **out) try
{
    if (!out) return fail("b2f_init: null out");
    *out = nullptr;
    GraphOpts g;
    {
        std::string err;
        if (!parse_graph_opts(graph_opts, g, err)) return fail("b2f_init: " + err);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("b2f_init: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("b2f_init: bad device ordinal");
    HIPCHK(hipSetDevice(device));
    const char *name = name_or_path ? name_or_path : "Ours-Soft-ft-KITTI";   // back2future.lua:98
    std::vector<float> flat;
    bool past = false;
    unsigned long long seed = 2;
    float gain = 1.f;
    if (parse_random(name, &past, &seed, &gain)) {
        g.past_flow = past;
        flat.resize((size_t)param_count(g));
        random_weights(seed, g, gain, flat.data());
    } else {
        std::string path = name;
        if (path == "Ours-Hard") path = "models/RoamingImages_H.t7";                      // :100-102
        else if (path == "Ours-Soft-ft-KITTI") path = "models/RoamingImages_H_KITTI_S.t7";   // :104-106
        else if (path == "Ours-Soft-ft-Sintel") path = "models/RoamingImages_H_Sintel_S.t7"; // :108-110
        if (ends_with(path, ".t7")) {
            // without graph options the shape (win, levels, skip) is read from the file; with them the file must be that graph
            std::string err;
            if (!load_t7_ex(path, g, !(graph_opts && *graph_opts), flat, err)) return fail("b2f_init: " + err);
            past = g.past_flow;
        } else if (ends_with(path, ".b2fw")) {
            FILE *f = fopen(path.c_str(), "rb");
            if (!f) return fail("b2f_init: cannot open " + path);
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            flat.resize((size_t)sz / 4);
            const size_t rd = fread(flat.data(), 4, flat.size(), f);
            fclose(f);
            if (rd != flat.size()) return fail("b2f_init: short read on " + path);
            GraphOpts gp = g, gn = g;
            gp.past_flow = true; gn.past_flow = false;
            if ((long long)flat.size() == param_count(gp)) past = true;
            else if ((long long)flat.size() == param_count(gn)) past = false;
            else return fail("b2f_init: " + path + " does not hold a parameter set of this graph (with or without past-flow decoders)");
        } else {
            return fail(std::string("b2f_init: unknown model '") + name + "'");
        }
    }
    b2f_ctx *c = new b2f_ctx();
    c->device = device;
    c->g = g;
    {   // defaults of the tuning options may come from the environment; read here once, never on the hot path
        // B2F_<OPTION NAME IN CAPITALS> seeds the option of that name (b2f_set_option keys); a value that is set but not a
        // number (B2F_PROFILE_LAYERS=yes, or empty) counts as 1
        auto env_int = [](const char *k, long long dflt) {
            const char *v = getenv(k);
            if (!v) return dflt;
            char *end = nullptr;
            const long long x = strtoll(v, &end, 10);
            return end == v ? 1ll : x;
        };
        c->wino4_min_pixels = (int)env_int("B2F_WINO4_MIN_PIXELS", c->wino4_min_pixels);
        c->wino_split_pixels = (int)env_int("B2F_WINO_SPLIT_PIXELS", c->wino_split_pixels);
        c->wino8 = (int)env_int("B2F_WINO8", c->wino8);
        c->adaptive_kernels = (int)env_int("B2F_ADAPTIVE_KERNELS", c->adaptive_kernels);
        c->corr_variant = (int)env_int("B2F_CORR_VARIANT", env_int("B2F_CORR_LAT", c->corr_variant));   // B2F_CORR_LAT: the round-1 name
        c->op_wino_split = (int)env_int("B2F_OP_WINO_SPLIT", c->op_wino_split);
        c->use_graph = (int)env_int("B2F_USE_GRAPH", c->use_graph);
        c->host_graph = (int)env_int("B2F_HOST_GRAPH", c->host_graph);
        c->profile = (int)env_int("B2F_PROFILE", c->profile);
        c->corr_ablate = (int)env_int("B2F_CORR_ABLATE", c->corr_ablate);
        c->profile_layers = (int)env_int("B2F_PROFILE_LAYERS", c->profile_layers);
        c->wino4_persistent = (int)env_int("B2F_WINO4_PERSISTENT", c->wino4_persistent);
        c->wino1d = (int)env_int("B2F_WINO1D", c->wino1d);
        c->wino6 = (int)env_int("B2F_WINO6", c->wino6);
        c->wino6_min_pixels = (int)env_int("B2F_WINO6_MIN_PIXELS", c->wino6_min_pixels);
        c->s2_loader = (int)env_int("B2F_S2_LOADER", c->s2_loader);
        c->s2_tile_groups = (int)env_int("B2F_S2_TILE_GROUPS", c->s2_tile_groups);
#if B2F_EXPERIMENTS
        c->wino4_split = (int)env_int("B2F_WINO4_SPLIT", c->wino4_split);
        c->wino4_hybrid = (int)env_int("B2F_WINO4_HYBRID", c->wino4_hybrid);
        c->wino2_split = (int)env_int("B2F_WINO2_SPLIT", c->wino2_split);
#endif
        if (!use_wino()) { c->bf16_conv = 0; c->bf16_direct = 0; }   // B2F_WINO=0: every layer on the direct fp32-MFMA kernel (bit-exact fmaf chains)
        c->bf16_direct = (int)env_int("B2F_BF16_DIRECT", c->bf16_direct);
        c->bf16_conv = (int)env_int("B2F_BF16_CONV", c->bf16_conv);
        c->s2_tiles_per_block = (int)env_int("B2F_S2_TILES_PER_BLOCK", c->s2_tiles_per_block);
        c->host_subbatch_pixels = env_int("B2F_HOST_SUBBATCH_PIXELS", c->host_subbatch_pixels);
        c->host_threads = (int)env_int("B2F_HOST_THREADS", c->host_threads);
        c->host_u8 = (int)env_int("B2F_HOST_U8", c->host_u8);
        c->host_ramp = (int)env_int("B2F_HOST_RAMP", c->host_ramp);
#if !B2F_EXPERIMENTS
        // the same rule b2f_set_option enforces: an experiment kernel that is not in this build is not accepted under its name (an A/B
        // run driven by the environment would otherwise time the default kernel under another label); get_option reports what runs
        if (c->corr_variant == 2 || c->corr_variant == 4 || c->corr_variant == 6 || c->corr_variant == 8) {
            fprintf(stderr, "b2f_init: B2F_CORR_VARIANT=%d is an experiment kernel (build with `python -m back2future_amd.build --experiments`): using the default\n", c->corr_variant);
            c->corr_variant = -1;
        }
        if (c->bf16_direct == 1) {
            fprintf(stderr, "b2f_init: B2F_BF16_DIRECT=1 is an experiment (build with `python -m back2future_amd.build --experiments`): using the default (2)\n");
            c->bf16_direct = 2;
        }
#endif
    }
    // a blocking stream: ordered with the legacy default stream like any such stream, so inputs that PyTorch (whose
    // default stream is the null stream) or hipMemcpy / hipMemset produced there are complete before our kernels read them
    if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) {
        delete c;
        return fail("b2f_init: hipStreamCreate failed");
    }
    if (install_weights(c, flat.data(), (long long)flat.size(), past) != 0) {
        b2f_destroy(c);
        return 1;
    }
    *out = c;
    return 0;
}
B2F_CATCH("b2f_init_ex")

void b2f_destroy(b2f_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graphs(c);
    drop_gen_out(c);
    for (ProfEvent &pe : c->prof_pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (HostSlot &hs : c->slot) {
        if (hs.dev) (void)hipFree(hs.dev);
        if (hs.pin) (void)hipHostFree(hs.pin);
        for (hipEvent_t e : {hs.ev_in, hs.ev_comp, hs.ev_out})
            if (e) (void)hipEventDestroy(e);
    }
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->arena) (void)hipFree(c->arena);
    if (c->wpk_dev) (void)hipFree(c->wpk_dev);
    if (c->w_dev) (void)hipFree(c->w_dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int b2f_info(const b2f_ctx *c, int *levels, int *win, int *past_flow, int *n_outputs, long long *n_params) try
{
    if (!c) return fail("b2f_info: null context");
    if (levels) *levels = c->g.levels;
    if (win) *win = c->g.win;
    if (past_flow) *past_flow = c->past_flow ? 1 : 0;
    if (n_outputs) *n_outputs = c->g.n_outputs();   // pwc.lua:459-489
    if (n_params) *n_params = c->nparams;
    return 0;
}
B2F_CATCH("b2f_info")

int b2f_set_weights(b2f_ctx *c, const float *host_flat, long long n) try
{
    if (!c || !host_flat) return fail("b2f_set_weights: null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    bool past;
    GraphOpts gp = c->g, gn = c->g;
    gp.past_flow = true; gn.past_flow = false;
    if (n == param_count(gp)) past = true;
    else if (n == param_count(gn)) past = false;
    else return fail("b2f_set_weights: n is neither the Hard nor the Soft parameter count of this graph");
    return install_weights(c, host_flat, n, past);
}
B2F_CATCH("b2f_set_weights")

int b2f_get_weights(b2f_ctx *c, float *host_flat, long long n) try
{
    if (!c || !host_flat || n != c->nparams) return fail("b2f_get_weights: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(host_flat, c->w_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_get_weights")

int b2f_weights_device(b2f_ctx *c, void **dev_ptr, long long *n) try
{
    if (!c || !dev_ptr) return fail("b2f_weights_device: null argument");
    *dev_ptr = c->w_dev;
    if (n) *n = c->nparams;
    return 0;
}
B2F_CATCH("b2f_weights_device")

int b2f_commit_weights(b2f_ctx *c) try
{
    if (!c) return fail("b2f_commit_weights: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    drop_graphs(c);   // pack_all may move wpk_dev
    std::vector<float> flat((size_t)c->nparams);
    HIPCHK(hipMemcpy(flat.data(), c->w_dev, flat.size() * sizeof(float), hipMemcpyDeviceToHost));
    return pack_all(c, flat.data());
}
B2F_CATCH("b2f_commit_weights")

int b2f_set_option(b2f_ctx *c, const char *key, int value) try
{
    if (!c || !key) return fail("b2f_set_option: null argument");
    if (!strcmp(key, "use_graph")) c->use_graph = value;
    else if (!strcmp(key, "host_graph")) c->host_graph = value;
    else if (!strcmp(key, "profile")) c->profile = value;
    else if (!strcmp(key, "profile_layers")) c->profile_layers = value;
    else if (!strcmp(key, "bf16_direct") || !strcmp(key, "bf16_conv") || !strcmp(key, "bf16_conv_min_pixels")) {
        // a different kernel mix: captured graphs hold the old one
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        if (!strcmp(key, "bf16_direct")) {
            if (value == 1 && !B2F_EXPERIMENTS) return fail("b2f_set_option: bf16_direct = 1 (the 16 -> 16 layer alone on the bf16 pipe) is an experiment: build with `python -m back2future_amd.build --experiments`");
            c->bf16_direct = value;
        } else if (!strcmp(key, "bf16_conv_min_pixels")) c->bf16_conv_min_pixels = value;
        else {
            const bool had = c->bf16_conv >= 2;
            c->bf16_conv = value;
            if (had != (value >= 2)) CHK(b2f_commit_weights(c));     // the F(4x4)-class layers carry the bf16 direct packing only while bf16_conv >= 2
        }
    }
    else if (!strcmp(key, "wino1d")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        const bool had = c->wino1d != 0;
        c->wino1d = value;
        if (had != (value != 0)) CHK(b2f_commit_weights(c));   // the packing exists only while the option reads it
    }
    else if (!strcmp(key, "wino6") || !strcmp(key, "wino6_min_pixels")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        if (key[5] == '_') c->wino6_min_pixels = value;
        else {
            const bool had = c->wino6 != 0;
            c->wino6 = value;
            if (had != (value != 0)) CHK(b2f_commit_weights(c));   // the packing exists only while the option reads it
        }
    }
#if !B2F_EXPERIMENTS
    else if (!strcmp(key, "wino2_split") || !strcmp(key, "wino4_split") || !strcmp(key, "wino4_hybrid")) {
        if (value != 0) return fail(std::string("b2f_set_option: ") + key + " selects an experiment kernel (tools/experiments/csrc): build with `python -m back2future_amd.build --experiments` and load libb2f_exp.so");
    }
#else
    else if (!strcmp(key, "wino2_split")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        const bool had = c->wino2_split != 0;
        c->wino2_split = value;
        if (had != (value != 0)) CHK(b2f_commit_weights(c));
    }
    else if (!strcmp(key, "wino4_split") || !strcmp(key, "wino4_hybrid")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        const bool had = c->wino4_split || c->wino4_hybrid;
        (key[6] == 's' ? c->wino4_split : c->wino4_hybrid) = value;
        if (had != (c->wino4_split || c->wino4_hybrid)) CHK(b2f_commit_weights(c));   // the split packing exists only while an option reads it
    }
#endif
    else if (!strcmp(key, "s2_tiles_per_block") || !strcmp(key, "wino4_persistent")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        (key[0] == 's' ? c->s2_tiles_per_block : c->wino4_persistent) = value;
    }
    else if (!strcmp(key, "s2_loader") || !strcmp(key, "s2_tile_groups")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        (key[3] == 'l' ? c->s2_loader : c->s2_tile_groups) = value;
    }
    else if (!strcmp(key, "wino_split_pixels") || !strcmp(key, "wino8")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        (key[4] == '8' ? c->wino8 : c->wino_split_pixels) = value;
    }
    else if (!strcmp(key, "wino4_min_pixels") || !strcmp(key, "adaptive_kernels")) {
        // a different kernel mix: captured graphs hold the old one
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        (key[0] == 'w' ? c->wino4_min_pixels : c->adaptive_kernels) = value;
    } else if (!strcmp(key, "corr_variant") || !strcmp(key, "corr_ablate")) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceSynchronize());
        drop_graphs(c);
        if (key[5] == 'v' && !B2F_EXPERIMENTS && (value == 2 || value == 4 || value == 6 || value == 8))
            return fail("b2f_set_option: corr_variant 2 / 4 / 6 / 8 are experiment kernels: build with `python -m back2future_amd.build --experiments`");
        (key[5] == 'v' ? c->corr_variant : c->corr_ablate) = value;
    } else if (!strcmp(key, "op_wino_split")) c->op_wino_split = value;
    else if (!strcmp(key, "host_subbatch_pixels")) c->host_subbatch_pixels = value > 0 ? value : (16ll << 20);
    else if (!strcmp(key, "host_threads")) c->host_threads = value;
    else if (!strcmp(key, "host_u8")) c->host_u8 = value;
    else if (!strcmp(key, "host_ramp")) c->host_ramp = value;
    else if (!strcmp(key, "debug_fail_next")) c->debug_fail_next = value;
    else return fail(std::string("b2f_set_option: unknown key ") + key);
    return 0;
}
B2F_CATCH("b2f_set_option")

int b2f_get_option(const b2f_ctx *c, const char *key, int *value) try
{
    if (!c || !key || !value) return fail("b2f_get_option: null argument");
    const std::string k(key);
    if (k == "use_graph") *value = c->use_graph;
    else if (k == "host_graph") *value = c->host_graph;
    else if (k == "profile") *value = c->profile;
    else if (k == "profile_layers") *value = c->profile_layers;
    else if (k == "s2_tiles_per_block") *value = c->s2_tiles_per_block;
    else if (k == "wino4_persistent") *value = c->wino4_persistent;
    else if (k == "experiments") *value = B2F_EXPERIMENTS;
    else if (k == "wino1d") *value = c->wino1d;
    else if (k == "wino6") *value = c->wino6;
    else if (k == "wino6_min_pixels") *value = c->wino6_min_pixels;
    else if (k == "s2_loader") *value = c->s2_loader;
    else if (k == "s2_tile_groups") *value = c->s2_tile_groups;
    else if (k == "wino4_split") *value = c->wino4_split;
    else if (k == "wino4_hybrid") *value = c->wino4_hybrid;
    else if (k == "wino2_split") *value = c->wino2_split;
    else if (k == "bf16_direct") *value = c->bf16_direct;
    else if (k == "bf16_conv") *value = c->bf16_conv;
    else if (k == "bf16_conv_min_pixels") *value = c->bf16_conv_min_pixels;
    else if (k == "wino4_min_pixels") *value = c->wino4_min_pixels;
    else if (k == "wino_split_pixels") *value = c->wino_split_pixels;
    else if (k == "wino8") *value = c->wino8;
    else if (k == "adaptive_kernels") *value = c->adaptive_kernels;
    else if (k == "corr_variant") *value = c->corr_variant;
    else if (k == "corr_ablate") *value = c->corr_ablate;
    else if (k == "op_wino_split") *value = c->op_wino_split;
    else if (k == "host_subbatch_pixels") *value = (int)std::min<long long>(c->host_subbatch_pixels, 0x7fffffff);
    else if (k == "host_threads") *value = c->host_threads;
    else if (k == "host_u8") *value = c->host_u8;
    else if (k == "host_ramp") *value = c->host_ramp;
    else if (k == "debug_fail_next") *value = c->debug_fail_next;
    else return fail("b2f_get_option: unknown key " + k);
    return 0;
}
B2F_CATCH("b2f_get_option")

int b2f_synchronize(b2f_ctx *c) try
{
    if (!c) return fail("b2f_synchronize: null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
B2F_CATCH("b2f_synchronize")

int b2f_profile_reset(b2f_ctx *c) try
{
    if (!c) return fail("b2f_profile_reset: null context");
    CHK(prof_collect(c));
    std::fill(c->prof_ms.begin(), c->prof_ms.end(), 0.0);
    std::fill(c->prof_n.begin(), c->prof_n.end(), 0);
    return 0;
}
B2F_CATCH("b2f_profile_reset")

int b2f_profile_read(b2f_ctx *c, char *names, double *total_ms, long long *launches, int cap, int *n) try
{
    if (!c || !n) return fail("b2f_profile_read: null argument");
    CHK(prof_collect(c));
    const int cnt = std::min<int>(cap, (int)c->prof_names.size());
    for (int i = 0; i < cnt; ++i) {
        if (names) { strncpy(names + 32 * i, c->prof_names[i].c_str(), 31); names[32 * i + 31] = 0; }
        if (total_ms) total_ms[i] = c->prof_ms[i];
        if (launches) launches[i] = c->prof_n[i];
    }
    *n = cnt;
    return 0;
}
B2F_CATCH("b2f_profile_read")

}  // extern "C"

// model:forward on device pointers.  With `graph` the ~150 launches of a (shape, pointers) combination are replayed
// from a hipGraph: the first call of a combination runs eagerly (and lets the kernels set their function
// attributes, which must not happen inside a capture), the second one captures, later ones only replay.  Worth
// ~0.5 ms per forward pass: 17 % of a single full-HD triplet, 2 % of a batch of 16.
int b2f::forward_device(b2f_ctx *c, const void *dev_in, int in_kind, int B, int H, int W, float *dev_flow, float *dev_occ,
                        float *dev_est3, hipStream_t s, bool graph)
{
    CHK(check_shape(B, H, W));
    HIPCHK(hipSetDevice(c->device));
    if (!c->g.shipped()) {
        // other graph shapes: the whole output table through the generic executor (b2f_graph.hip), then est[1] / the
        // finest occlusion map / est[3] into the caller's buffers.  Asynchronous on the stream like the tuned path, no hipGraph: a correctness path.
        const int n = c->g.n_outputs(), per = c->past_flow ? 5 : 4, lst = c->g.l_st();
        // the output table lives in the context and is re-allocated only when the shape changes (a hipMalloc / hipFree pair per
        // call is device-synchronising and stalled the upload / download overlap of the host pipeline: ADVICE r2)
        if ((int)c->gen_out.size() != n || c->genB != B || c->genH != H || c->genW != W) {
            HIPCHK(hipDeviceSynchronize());
            drop_gen_out(c);
            c->gen_out.assign((size_t)n, nullptr);
            for (int i = 0; i < n; ++i) {
                const int l = lst + i / per, j = i % per;
                const size_t cnt = (size_t)B * (H >> (l - lst)) * (W >> (l - lst)) * ((j >= per - 2) ? 3 : 2);
                if (hipMalloc(&c->gen_out[(size_t)i], cnt * sizeof(float)) != hipSuccess) {
                    drop_gen_out(c);
                    return fail("b2f_forward_device: out of device memory");
                }
            }
            c->genB = B; c->genH = H; c->genW = W;
        }
        std::vector<float *> &dev = c->gen_out;
        CHK(graph_forward(c, s, false, dev_in, in_kind, B, H, W, dev.data()));
        const size_t n2 = (size_t)B * 2 * H * W * sizeof(float);
        if (dev_flow) HIPCHK(hipMemcpyAsync(dev_flow, dev[0], n2, hipMemcpyDeviceToDevice, s));
        if (dev_occ) HIPCHK(hipMemcpyAsync(dev_occ, dev[(size_t)(c->past_flow ? 2 : 1)], n2, hipMemcpyDeviceToDevice, s));
        if (dev_est3) HIPCHK(hipMemcpyAsync(dev_est3, dev[2], c->past_flow ? n2 : n2 / 2 * 3, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    const Plan P = make_plan(B, H, W, false, c->past_flow);
    CHK(ensure_workspace(c, P));
    Outs O;
    O.flow = dev_flow; O.occ = dev_occ; O.est3 = dev_est3;
    if (graph && !c->profile) {
        if (c->graphs.size() > 256) {   // callers that keep changing pointers: start over (replays may still be in flight)
            HIPCHK(hipDeviceSynchronize());
            drop_graphs(c);
        }
        const int req = c->req_batch > 0 ? c->req_batch : B;
        const GraphKey key = {dev_in, dev_flow, dev_occ, dev_est3, in_kind, B, H, W, (c->adaptive_kernels > 0 || (c->adaptive_kernels < 0 && req == 1)) ? 1 : 0};
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            c->graphs.emplace(key, nullptr);
            return forward_impl(c, s, false, dev_in, in_kind, P, O);
        }
        if (!it->second) {
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rc = forward_impl(c, s, true, dev_in, in_kind, P, O);
            const hipError_t e = hipStreamEndCapture(s, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            HIPCHK(e);
            hipGraphExec_t ge = nullptr;
            const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            HIPCHK(ei);
            it->second = ge;
        }
        HIPCHK(hipGraphLaunch(it->second, s));
        return 0;
    }
    return forward_impl(c, s, false, dev_in, in_kind, P, O);
}

extern "C" {

int b2f_forward_device(b2f_ctx *c, const void *dev_in, int in_kind, int B, int H, int W, float *dev_flow,
                       float *dev_occ, float *dev_est3, void *stream) try
{
    if (!c || !dev_in) return fail("b2f_forward_device: null argument");
    // the glue kernels read / write these planes with 16-byte vector accesses (include/b2f.h states the requirement)
    if (((uintptr_t)dev_in | (uintptr_t)dev_flow | (uintptr_t)dev_occ | (uintptr_t)dev_est3) & 15)
        return fail("b2f_forward_device: device buffers must be 16-byte aligned");
    return forward_device(c, dev_in, in_kind, B, H, W, dev_flow, dev_occ, dev_est3, stream ? (hipStream_t)stream : c->stream,
                          c->use_graph != 0);
}
B2F_CATCH("b2f_forward_device")

int b2f_output_shapes(const b2f_ctx *c, int H, int W, int *ch, int *oh, int *ow, int cap) try
{
    if (!c) return fail("b2f_output_shapes: null context");
    int no = 0;
    for (int l = c->g.l_st(); l <= c->g.levels; ++l) {
        const int per = c->past_flow ? 5 : 4;
        for (int j = 0; j < per; ++j) {
            if (no >= cap) return fail("b2f_output_shapes: cap too small");
            ch[no] = (j >= per - 2) ? 3 : 2;
            oh[no] = H >> (l - c->g.l_st());
            ow[no] = W >> (l - c->g.l_st());
            ++no;
        }
    }
    return 0;
}
B2F_CATCH("b2f_output_shapes")

int b2f_forward(b2f_ctx *c, const float *x, int B, int H, int W, float **outs, int n_outs) try
{
    if (!c || !x || !outs) return fail("b2f_forward: null argument");
    const bool shipped = c->g.shipped();
    if (shipped) CHK(check_shape(B, H, W));
    else if (B <= 0 || H <= 0 || W <= 0 || H % (1 << (c->g.levels - 1)) || W % (1 << (c->g.levels - 1)))
        return fail("b2f_forward: H and W must be positive multiples of 2^(levels - 1)");
    HIPCHK(hipSetDevice(c->device));
    const int per = c->past_flow ? 5 : 4, lst = c->g.l_st(), lev = c->g.levels;
    if (n_outs != c->g.n_outputs()) return fail("b2f_forward: n_outs must be (levels - skip) x 4 (5 with past-flow decoders): 20 / 25 for the shipped models");
    // arena first (it may be re-allocated), then the per-output device buffers
    Plan P;
    if (shipped) {
        P = make_plan(B, H, W, true, c->past_flow);
        CHK(ensure_workspace(c, P));
    }
    std::vector<float *> dev(n_outs, nullptr);
    std::vector<size_t> cnt(n_outs, 0);
    float *d_in = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        for (float *p : dev) if (p) (void)hipFree(p);
        if (d_in) (void)hipFree(d_in);
    };
    Outs O;
    int no = 0;
    for (int l = lst; l <= lev && !rc; ++l) {
        const size_t px = (size_t)B * (H >> (l - lst)) * (W >> (l - lst));
        for (int j = 0; j < per; ++j) {
            const int ch = (j >= per - 2) ? 3 : 2;
            cnt[no] = px * ch;
            if (hipMalloc(&dev[no], cnt[no] * sizeof(float)) != hipSuccess) { rc = fail("b2f_forward: out of device memory"); break; }
            ++no;
        }
        if (rc) break;
        int base = (l - lst) * per, j = 0;
        O.t_ufs[l] = dev[base + j++];
        if (c->past_flow) O.t_ubfs[l] = dev[base + j++];
        O.t_occ[l] = dev[base + j++];
        O.t_iw1[l] = dev[base + j++];
        O.t_iw3[l] = dev[base + j++];
    }
    const size_t nin = (size_t)B * 9 * H * W;
    if (!rc && hipMalloc(&d_in, nin * sizeof(float)) != hipSuccess) rc = fail("b2f_forward: out of device memory");
    if (!rc && hipMemcpyAsync(d_in, x, nin * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail("b2f_forward: H2D copy failed");
    if (!rc) rc = shipped ? forward_impl(c, c->stream, false, d_in, B2F_IN_NORMALIZED, P, O)
                          : graph_forward(c, c->stream, false, d_in, B2F_IN_NORMALIZED, B, H, W, dev.data());
    for (int i = 0; i < n_outs && !rc; ++i)
        if (hipMemcpyAsync(outs[i], dev[i], cnt[i] * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail("b2f_forward: D2H copy failed");
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(std::string("b2f_forward: ") + hipGetErrorString(hipGetLastError()));
    cleanup();
    return rc;
}
B2F_CATCH("b2f_forward")

// ---- op-level entry points (host pointers; reference module layouts) -----------------------
namespace {
struct DevBuf {
    float *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float))); return 0; }
};
}  // namespace

int b2f_op_warp_bhwd(b2f_ctx *c, const float *img, const float *grid, int B, int ih, int iw, int C, int gh,
                     int gw, float *out) try
{
    if (!c || !img || !grid || !out) return fail("b2f_op_warp_bhwd: null argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf di, dg, dout;
    const size_t ni = (size_t)B * ih * iw * C, ng = (size_t)B * gh * gw * 2, no = (size_t)B * gh * gw * C;
    CHK(di.alloc(ni)); CHK(dg.alloc(ng)); CHK(dout.alloc(no));
    HIPCHK(hipMemcpy(di.p, img, ni * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dg.p, grid, ng * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_warp_nhwc(di.p, (long)((size_t)ih * iw * C), C, C, ih, iw, dg.p, 1.0f, B, gh, gw, dout.p, C, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, dout.p, no * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_warp_bhwd")

int b2f_op_image_scale(b2f_ctx *c, const float *src, int C, int Hs, int Ws, int normalize, float *dst, int Hd, int Wd) try
{
    if (!c || !src || !dst) return fail("b2f_op_image_scale: null argument");
    if (C <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return fail("b2f_op_image_scale: bad shape");
    HIPCHK(hipSetDevice(c->device));
    DevBuf ds, dt, dd;
    const size_t ns = (size_t)C * Hs * Ws, nt = (size_t)C * Hs * Wd, nd = (size_t)C * Hd * Wd;
    CHK(ds.alloc(ns)); CHK(dt.alloc(nt)); CHK(dd.alloc(nd));
    HIPCHK(hipMemcpy(ds.p, src, ns * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_image_scale(ds.p, normalize, C, Hs, Ws, dt.p, dd.p, Hd, Wd, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(dst, dd.p, nd * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_image_scale")

int b2f_op_upsample_flow2x(b2f_ctx *c, const float *x, int B, int h, int w, float *y) try
{
    if (!c || !x || !y) return fail("b2f_op_upsample_flow2x: null argument");
    HIPCHK(hipSetDevice(c->device));
    DevBuf dp, dn, dy;
    const size_t n = (size_t)B * 2 * h * w;
    CHK(dp.alloc(n)); CHK(dn.alloc(n)); CHK(dy.alloc(4 * n));
    HIPCHK(hipMemcpy(dp.p, x, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_planar_to_nhwc(dp.p, 2, B, h, w, dn.p, 2, c->stream));
    HIPCHK(launch_upsample_flow2x_planar(dn.p, 2, B, h, w, dy.p, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(y, dy.p, 4 * n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_upsample_flow2x")

int b2f_op_warp_costvol(b2f_ctx *c, const float *ref, const float *nbr_future, const float *nbr_past,
                        const float *flow, float k, int B, int C, int h, int w, float *out) try
{
    if (!c || !ref || !nbr_future || !nbr_past || !out) return fail("b2f_op_warp_costvol: null argument");
    HIPCHK(hipSetDevice(c->device));
    const int Cp = (C + 7) / 8 * 8;   // the fused kernel walks channels in chunks of 8; zero channels add 0
    const size_t hw = (size_t)h * w, nplanar = (size_t)B * C * hw, nn = (size_t)B * hw * Cp;
    DevBuf dpl, dr, df, dpa, dfl_pl, dfl, dcv;
    CHK(dpl.alloc(nplanar)); CHK(dr.alloc(nn)); CHK(df.alloc(nn)); CHK(dpa.alloc(nn));
    CHK(dcv.alloc((size_t)B * hw * kCvRec));
    const float *srcs[3] = {ref, nbr_future, nbr_past};
    float *dsts[3] = {dr.p, df.p, dpa.p};
    for (int i = 0; i < 3; ++i) {
        HIPCHK(hipMemcpyAsync(dpl.p, srcs[i], nplanar * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_planar_to_nhwc(dpl.p, C, B, h, w, dsts[i], Cp, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (flow) {
        CHK(dfl_pl.alloc((size_t)B * 2 * hw)); CHK(dfl.alloc((size_t)B * 2 * hw));
        HIPCHK(hipMemcpy(dfl_pl.p, flow, (size_t)B * 2 * hw * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(launch_planar_to_nhwc(dfl_pl.p, 2, B, h, w, dfl.p, 2, c->stream));
    }
    CorrLaunch cl;
    cl.ref = dr.p; cl.nbr_fut = df.p; cl.nbr_past = dpa.p;
    cl.img_stride = (long)(hw * Cp); cl.chunk_stride = 8; cl.pix_stride = Cp;   // NHWC expressed with strides
    cl.flow = flow ? dfl.p : nullptr;
    cl.flow_b = nullptr;
    cl.k = k; cl.out = dcv.p;
    cl.out_img_stride = (long)(hw * kCvRec); cl.out_chunk_stride = 8; cl.out_pix_stride = kCvRec;
    cl.B = B; cl.C = Cp; cl.h = h; cl.w = w;
    cl.variant = c->corr_variant;
    HIPCHK(launch_warp_costvol(cl, c->stream));
    std::vector<float> rec((size_t)B * hw * kCvRec);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(rec.data(), dcv.p, rec.size() * sizeof(float), hipMemcpyDeviceToHost));
    // record slots -> the reference's JoinTable order {fwd 81, bwd 81}; the kernel divided by Cp,
    // CostVolMulti.lua:100 divides by N = C
    const float fix = (Cp != C) ? (float)Cp / (float)C : 1.f;
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < 2; ++d)
            for (int ch = 0; ch < 81; ++ch) {
                const int slot = cv_slot(d, ch);
                float *dst = out + ((size_t)b * kND + d * 81 + ch) * hw;
                const float *src = rec.data() + (size_t)b * hw * kCvRec + slot;
                for (size_t i = 0; i < hw; ++i) dst[i] = src[i * kCvRec] * fix;
            }
    return 0;
}
B2F_CATCH("b2f_op_warp_costvol")

int b2f_op_costvol(b2f_ctx *c, const float *ref, const float *frm, int B, int C, int h, int w, int win, int fwd,
                   float *out) try
{
    if (!c || !ref || !frm || !out) return fail("b2f_op_costvol: null argument");
    if (win < 1 || win % 2 == 0) return fail("b2f_op_costvol: win must be odd");
    HIPCHK(hipSetDevice(c->device));
    const size_t hw = (size_t)h * w;
    if (win == kWin && C % 8 == 0) {
        // the shipped 9x9 window goes through the fused kernel (no warp): one half of its record
        std::vector<float> both((size_t)B * kND * hw);
        CHK(b2f_op_warp_costvol(c, ref, frm, frm, nullptr, 0.f, B, C, h, w, both.data()));
        for (int b = 0; b < B; ++b)
            memcpy(out + (size_t)b * 81 * hw, both.data() + ((size_t)b * kND + (fwd ? 0 : 81)) * hw, 81 * hw * sizeof(float));
        return 0;
    }
    DevBuf dpl, dr, df, dcv, dout;
    const size_t nin = (size_t)B * C * hw, nout = (size_t)B * win * win * hw;
    CHK(dpl.alloc(nin)); CHK(dr.alloc(nin)); CHK(df.alloc(nin)); CHK(dcv.alloc(nout)); CHK(dout.alloc(nout));
    HIPCHK(hipMemcpy(dpl.p, ref, nin * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_planar_to_nhwc(dpl.p, C, B, h, w, dr.p, C, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(dpl.p, frm, nin * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_planar_to_nhwc(dpl.p, C, B, h, w, df.p, C, c->stream));
    HIPCHK(launch_costvol_generic(dr.p, df.p, B, C, h, w, win, fwd, dcv.p, c->stream));
    HIPCHK(launch_nhwc_to_planar(dcv.p, win * win, win * win, B, h, w, dout.p, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, dout.p, nout * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_costvol")

int b2f_op_conv3x3(b2f_ctx *c, const float *x, int B, int Ci, int H, int W, const float *wt, const float *bias,
                   int Co, int stride, int leaky, float *y) try
{
    if (!c || !x || !wt || !bias || !y) return fail("b2f_op_conv3x3: null argument");
    if (stride != 1 && stride != 2) return fail("b2f_op_conv3x3: stride must be 1 or 2");
    HIPCHK(hipSetDevice(c->device));
    const int chunks = (Ci + kCK - 1) / kCK, Cp = chunks * kCK;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int nt, nblk;
    int wino = (stride == 1 && use_wino() && Ci == 16 && Co == 16) ? 3 : (stride == 2 && use_wino() && Ci == 16 && Co == 32) ? 5 : stride == 1 ? wino_mode(Co) : 0;
    // tests: option op_wino_split = 1 runs F(4x4)-eligible layers on the F(2x2) kernel, one block per 32-output N tile
    const bool op_split = c->op_wino_split && wino == 4 && Co > 32;
    if (op_split) wino = 2;
    if (wino == 4) { nt = 2; nblk = wino4_nblk(Co); }
    else if (wino == 1 || wino == 3 || wino == 5) { nt = 1; nblk = 1; }
    else if (wino == 2) wino_choose_tiles(Co, &nt, &nblk);
    else conv_choose_tiles(Co, &nt, &nblk);
    std::vector<float> wpk(wino == 4 ? wino4_wpk_floats(chunks, nblk) : wino == 1 ? narrow2_wpk_floats(chunks)
                           : wino == 3 ? c16_wpk_floats() : wino == 5 ? c16s2_wpk_floats() : wino == 2 ? wino_wpk_floats(chunks, nt, nblk) : conv_wpk_floats(chunks, nt, nblk)),
        bpk((size_t)nblk * nt * 32);
    if (wino == 4) wino4_pack_weights(wt, bias, Co, Ci, nullptr, chunks, nblk, wpk.data(), bpk.data());
    else if (wino == 1) narrow2_pack_weights(wt, bias, Ci, nullptr, chunks, wpk.data(), bpk.data());
    else if (wino == 3) c16_pack_weights(wt, bias, Ci, nullptr, wpk.data(), bpk.data());
    else if (wino == 5) c16s2_pack_weights(wt, bias, Ci, nullptr, wpk.data(), bpk.data());
    else if (wino == 2) wino_pack_weights(wt, bias, Co, Ci, nullptr, chunks, nt, nblk, wpk.data(), bpk.data());
    else conv_pack_weights(wt, bias, Co, Ci, nullptr, chunks, nt, nblk, wpk.data(), bpk.data());
    DevBuf dpl, dx, dw, db, dy, dyp, dws, dws2;
    const size_t nx = (size_t)B * Ci * H * W, nxp = (size_t)B * H * W * Cp, ny = (size_t)B * Co * Ho * Wo;
    CHK(dpl.alloc(nx)); CHK(dx.alloc(nxp)); CHK(dw.alloc(wpk.size())); CHK(db.alloc(bpk.size())); CHK(dy.alloc(ny)); CHK(dyp.alloc(ny));
#if B2F_EXPERIMENTS
    if (wino == 4 && (c->wino4_split || c->wino4_hybrid)) {
        std::vector<float> wps(wino4s_wpk_floats(chunks, nblk));
        wino4s_pack_weights(wt, Co, Ci, nullptr, chunks, nblk, wps.data());
        CHK(dws.alloc(wps.size()));
        HIPCHK(hipMemcpy(dws.p, wps.data(), wps.size() * sizeof(float), hipMemcpyHostToDevice));
    }
#endif
    HIPCHK(hipMemcpy(dpl.p, x, nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw.p, wpk.data(), wpk.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, bpk.data(), bpk.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_planar_to_nhwc(dpl.p, Ci, B, H, W, dx.p, Cp, c->stream));
    ConvLaunch L;
    L.nseg = 1;
    L.seg[0] = {dx.p, (long)((size_t)H * W * Cp), 8, Cp, chunks};
    L.seg[1] = L.seg[0];
    L.seg[1].nchunks = 0;
    L.wpk = dw.p; L.bias = db.p; L.out = dy.p;
    L.out_img_stride = (long)((size_t)Ho * Wo * Co); L.out_chunk_stride = 8; L.out_pix_stride = Co; L.cout = Co;
    L.nt = nt; L.nblk = nblk; L.H = H; L.W = W; L.Ho = Ho; L.Wo = Wo; L.stride = stride; L.nimg = B; L.leaky = leaky;
    L.nsplit = op_split ? 1 : 0;
    L.nb0 = 0; L.trace = nullptr;
    L.w4_persist = c->wino4_persistent;
    L.w8 = c->wino8;
    L.wpk_split = dws.p;
    L.w4_hybrid = c->wino4_hybrid;
    L.bf16_direct = c->bf16_direct;
    DevBuf dw5, db5;
    const bool bf6_op = (Co & 3) == 0 && ((wino == 0 && c->bf16_conv) || (wino == 4 && stride == 1 && H * W >= c->bf16_conv_min_pixels && ((c->bf16_conv == 2 && Co <= 32) || c->bf16_conv >= 3)));
    if (bf6_op) {
        std::vector<float> w5(convb_wpk_floats(chunks, Co)), b5((size_t)convb_nblk(Co) * 64);
        convb_pack_weights(wt, bias, Co, Ci, nullptr, chunks, w5.data(), b5.data());
        CHK(dw5.alloc(w5.size())); CHK(db5.alloc(b5.size()));
        HIPCHK(hipMemcpy(dw5.p, w5.data(), w5.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db5.p, b5.data(), b5.size() * sizeof(float), hipMemcpyHostToDevice));
        L.wpk_bf6 = dw5.p; L.bias_bf6 = db5.p;
    }
    DevBuf dw7, db7;
    bool s2l_op = false;
    if (wino == 0 && stride == 2 && c->bf16_conv && c->s2_loader && (Co & 3) == 0 && Co <= 256 && (c->s2_loader >= 2 || chunks >= 8)) {
        std::vector<float> w7(s2b_wpk_floats(chunks, Co)), b7((size_t)s2b_ntiles(Co) * 32);
        s2b_pack_weights(wt, bias, Co, Ci, nullptr, chunks, w7.data(), b7.data());
        CHK(dw7.alloc(w7.size())); CHK(db7.alloc(b7.size()));
        HIPCHK(hipMemcpy(dw7.p, w7.data(), w7.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db7.p, b7.data(), b7.size() * sizeof(float), hipMemcpyHostToDevice));
        L.wpk_s2b = dw7.p; L.bias_s2b = db7.p;
        s2l_op = s2b_supported(L);
        if (s2l_op) L.nsplit = c->s2_tile_groups ? 0 : -1;
    }
    DevBuf dw6, db6;
    bool w1d_op = false;
    if (wino == 4 && stride == 1 && c->wino1d && !bf6_op) {
        std::vector<float> w6(w1b_wpk_floats(chunks, Co)), b6((size_t)w1b_nblk(Co) * 64);
        w1b_pack_weights(wt, bias, Co, Ci, nullptr, chunks, w6.data(), b6.data());
        CHK(dw6.alloc(w6.size())); CHK(db6.alloc(b6.size()));
        HIPCHK(hipMemcpy(dw6.p, w6.data(), w6.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db6.p, b6.data(), b6.size() * sizeof(float), hipMemcpyHostToDevice));
        L.wpk_w1b = dw6.p; L.bias_w1b = db6.p;
        L.w1b_nblk = c->wino1d >= 2 ? w1b_nblk(Co) : Co / 64 + (Co % 64 > 32 ? 1 : 0);
        w1d_op = w1b_supported(L) && L.w1b_nblk > 0;
    }
    DevBuf dw8, db8;
    bool w6_op = false;
    if (wino == 4 && stride == 1 && c->wino6 && !bf6_op && !w1d_op && H * W >= c->wino6_min_pixels) {
        std::vector<float> w8(wino6_wpk_floats(chunks, Co)), b8((size_t)wino6_nblk(Co) * 64);
        wino6_pack_weights(wt, bias, Co, Ci, nullptr, chunks, w8.data(), b8.data());
        CHK(dw8.alloc(w8.size())); CHK(db8.alloc(b8.size()));
        HIPCHK(hipMemcpy(dw8.p, w8.data(), w8.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db8.p, b8.data(), b8.size() * sizeof(float), hipMemcpyHostToDevice));
        L.wpk_w6 = dw8.p; L.bias_w6 = db8.p;
        w6_op = wino6_supported(L);
    }
#if B2F_EXPERIMENTS
    if (wino == 4 && c->wino2_split) {
        std::vector<float> wps(wino2s_wpk_floats(chunks, nblk));
        wino2s_pack_weights(wt, Co, Ci, nullptr, chunks, nblk, wps.data());
        CHK(dws2.alloc(wps.size()));
        HIPCHK(hipMemcpy(dws2.p, wps.data(), wps.size() * sizeof(float), hipMemcpyHostToDevice));
        L.wpk_split2 = dws2.p;
    }
#endif
    if (s2l_op) HIPCHK(launch_conv3x3_s2b(L, c->stream));
    else if (bf6_op && convb_supported(L)) HIPCHK(launch_conv3x3_bf6(L, c->stream));
    else if (w1d_op) {
        HIPCHK(launch_conv3x3_w1b(L, c->stream));
        if (L.w1b_nblk < w1b_nblk(Co)) HIPCHK(launch_conv3x3_wino4_rem(L, c->stream));
    }
    else if (w6_op) HIPCHK(launch_conv3x3_wino6(L, c->stream));
    else if (wino == 4) HIPCHK(launch_conv3x3_wino4(L, c->stream));
    else if (wino == 1) HIPCHK(launch_conv_narrow2(L, c->stream));
    else if (wino == 3) HIPCHK(launch_conv3x3_c16(L, c->stream));
    else if (wino == 5) HIPCHK(launch_conv3x3_c16s2(L, c->stream));
    else if (wino == 2) HIPCHK(launch_conv3x3_wino(L, c->stream));
    else HIPCHK(launch_conv3x3(L, c->stream));
    HIPCHK(launch_nhwc_to_planar(dy.p, Co, Co, B, Ho, Wo, dyp.p, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(y, dyp.p, ny * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_conv3x3")

int b2f_op_conv_head16(b2f_ctx *c, const float *x, int B, int H, int W, const float *w1, const float *b1, const float *w2,
                       const float *b2, float *y) try
{
    if (!c || !x || !w1 || !b1 || !w2 || !b2 || !y) return fail("b2f_op_conv_head16: null argument");
    if (B < 1 || H < 1 || W < 1) return fail("b2f_op_conv_head16: bad size");
    HIPCHK(hipSetDevice(c->device));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    std::vector<float> p1(c16_wpk_floats()), pb1(32), p2(c16s2_wpk_floats()), pb2(32);
    c16_pack_weights(w1, b1, 16, nullptr, p1.data(), pb1.data());
    c16s2_pack_weights(w2, b2, 16, nullptr, p2.data(), pb2.data());
    DevBuf dpl, dx, dw1, db1, dw2, db2, dy, dyp;
    const size_t nx = (size_t)B * 16 * H * W, ny = (size_t)B * 32 * Ho * Wo;
    CHK(dpl.alloc(nx)); CHK(dx.alloc(nx)); CHK(dw1.alloc(p1.size())); CHK(db1.alloc(32)); CHK(dw2.alloc(p2.size())); CHK(db2.alloc(32));
    CHK(dy.alloc(ny)); CHK(dyp.alloc(ny));
    HIPCHK(hipMemcpy(dpl.p, x, nx * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw1.p, p1.data(), p1.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db1.p, pb1.data(), 32 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw2.p, p2.data(), p2.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db2.p, pb2.data(), 32 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(launch_planar_to_nhwc(dpl.p, 16, B, H, W, dx.p, 16, c->stream));
    HeadLaunch hl;
    hl.in = dx.p; hl.in_img_stride = (long)((size_t)H * W * 16); hl.in_chunk_stride = 8; hl.in_pix_stride = 16;
    hl.H1 = H; hl.W1 = W;
    hl.w1 = dw1.p; hl.b1 = db1.p; hl.w2 = dw2.p; hl.b2 = db2.p;
    hl.out = dy.p; hl.out_img_stride = (long)((size_t)Ho * Wo * 32); hl.out_chunk_stride = 8; hl.out_pix_stride = 32;
    hl.Ho = Ho; hl.Wo = Wo; hl.nimg = B;
    if (!head16_supported(hl)) return fail("b2f_op_conv_head16: unsupported shape");
    HIPCHK(launch_conv_head16(hl, c->stream));
    HIPCHK(launch_nhwc_to_planar(dy.p, 32, 32, B, Ho, Wo, dyp.p, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(y, dyp.p, ny * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_conv_head16")

}  // extern "C"
