// conv3x3 as an fp32 MFMA implicit GEMM for gfx950 (MI355X).
//
// Replaces nn.SpatialConvolution(Ci,Co,3,3,s,s,1,1) [+ nn.LeakyReLU(0.2,true)] of
// /root/reference/models/pwc.lua:58-65 (convUnit) and :76-85 (decoder), which the
// reference runs through cuDNN (model.lua:63-65).  This is the MFMA-bound ~90 % of the
// FLOPs of computeFlow.
//
// GEMM view: M = output pixels, N = output channels, K = 9 taps x Cin.
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles / 4096 FLOP.
//   * block = NW waves (8, or 4 for small maps); block tile = NW*32 pixels (TH x TW) x NT*32
//     channels; wave w owns one 32-pixel M tile and all NT N tiles (NT <= 4 -> <= 64 acc VGPRs).
//   * K is walked in chunks of 8 input channels.  Per chunk the block stages in LDS
//       A: the input patch with halo, im2col done by address arithmetic at read time,
//          laid out [k4 = 2][patch pixel] of float4 so that a half-wave reads 32
//          neighbouring pixels x 4 channels with one conflict-free ds_read_b128;
//       B: the pre-packed weight slab [tap 9][k4 2][NT*32 cout] of float4 (a straight
//          linear copy of the packed global layout).
//     Lane l = (n|m = l & 31, half = l >> 5).  For the 32x32x2 MFMA a lane supplies
//     A[m][k = half] and B[k = half][n]; half h holds channels 4h..4h+3 of the chunk, so
//     one b128 per operand feeds four consecutive MFMAs (k-pair j = {j, 4 + j}).
//   * two LDS buffers, ONE barrier per chunk: chunk c+1's global loads are issued before the
//     MFMAs of chunk c, written to the other buffer right after them, then the barrier; the
//     operands of tap t+1 are read from LDS before the 4*NT MFMAs of tap t.
//   * activations are chunk-planar ([image][C/8][h][w][8], b2f_internal.h): the 8-channel
//     chunk of a patch row is one contiguous run, so the A loads are whole cache lines.
//   * input may come from up to two K segments (pointer, strides, #chunks): the decoder's first
//     layer reads {cs[ref][l], cost-volume record} without a JoinTable copy (pwc.lua:308,334).
//   * epilogue: accumulators start at the bias, LeakyReLU fused, chunk-planar store (8 lanes
//     write the 32-B chunk of one pixel; the four stores of 4 neighbouring pixels fill a line).
#include "b2f_internal.h"

#include <cstdlib>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: staging arrays of it stay in VGPRs

template <int S, int TW, int NW>
struct ConvGeom {
    static constexpr int NTHR = NW * 64;
    static constexpr int TH = NW * 32 / TW;
    static constexpr int PH = (TH - 1) * S + 3;
    static constexpr int PW = (TW - 1) * S + 3;
    static constexpr int NPIX = PH * PW;
    static constexpr int A_F4 = 2 * NPIX;                 // float4 slots of the A patch
    static constexpr int A_PER_THREAD = (A_F4 + NTHR - 1) / NTHR;
};

template <int S, int NT, int TW, int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_mfma(const ConvLaunch p)
{
    using G = ConvGeom<S, TW, NW>;
    constexpr int NTHR = G::NTHR;
    constexpr int NPIX = G::NPIX, PW = G::PW;
    constexpr int NTOT = NT * 32;
    constexpr int B_F4 = 9 * 2 * NTOT;
    constexpr int B_PER_THREAD = (B_F4 + NTHR - 1) / NTHR;
    constexpr int A_PER_THREAD = G::A_PER_THREAD;
    constexpr int BUF_F4 = G::A_F4 + B_F4;                // one LDS buffer (A patch + B slab)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *lds = reinterpret_cast<f32x4 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, half = lane >> 5;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + G::TH - 1) / G::TH;
    const int ntiles_all = tiles_x * tiles_y * p.nimg;
    // A block walks `tpb` consecutive tiles (p.tiles_per_block; 1 = one tile per block).  With more than one the
    // (tile, chunk) pairs form ONE software pipeline: the loads of the next tile's first chunk are in flight under the
    // last MFMAs and the output stores of the current tile -- a layer with two K chunks (16 -> 32, stride 2) then runs
    // like a deep one instead of paying two memory round trips, a prologue and a store tail per 72 MFMAs.
    const int tpb = p.tiles_per_block > 1 ? p.tiles_per_block : 1;
    const int tile0 = blockIdx.x * tpb;
    const int my_tiles = min(tpb, ntiles_all - tile0);
    const int nb = blockIdx.y;
    int img = 0, ox0 = 0, oy0 = 0;               // tile of the item whose loads are issued next
    int st_img = 0, st_ox0 = 0, st_oy0 = 0;       // tile of the item being computed / stored

    // ---- per-thread staging coordinates for the A patch (fixed over the chunks of a tile) ----
    unsigned a_off0[A_PER_THREAD], a_off1[A_PER_THREAD];   // byte offset inside the (image, chunk) plane, per K segment
    bool a_ok[A_PER_THREAD];    // false: halo outside the image / beyond the patch -> zero fill
    int a_lds[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int idx = tid + i * NTHR;
        a_lds[i] = (idx < G::A_F4) ? ((idx & 1) * NPIX + (idx >> 1)) : -1;
    }
    __amdgpu_buffer_rsrc_t a_rsrc0, a_rsrc1;
    auto enter_tile = [&](int t) {
        int bid = t;
        const int tx_i = bid % tiles_x;
        bid /= tiles_x;
        const int ty_i = bid % tiles_y;
        img = bid / tiles_y;
        ox0 = tx_i * TW; oy0 = ty_i * G::TH;
        const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1;
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            const int idx = tid + i * NTHR;
            const int pix = idx >> 1;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = iy0 + py, gx = ix0 + px;
            a_ok[i] = (idx < G::A_F4) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned gp = a_ok[i] ? (unsigned)(gy * p.W + gx) : 0u;   // unconditional load from a valid address, no branch
            a_off0[i] = (gp * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
            a_off1[i] = (gp * (unsigned)p.seg[1].pix_stride + (tid & 1) * 4) * 4u;
        }
        // buffer loads: a scalar 128-bit resource (per K segment, based at this image), a scalar chunk offset and one
        // 32-bit lane offset -- no per-load address arithmetic on the VALU
        a_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
        a_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
    };
    enter_tile(tile0);
    st_img = img; st_ox0 = ox0; st_oy0 = oy0;

    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    // 16-byte output stores need 4-aligned strides and channel count
    const bool vec_ok = ((p.out_pix_stride | (int)p.out_chunk_stride | p.cout) & 3) == 0;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.wpk + (size_t)nb * nchunks * B_F4 * 4), 0, 0x7fffffff, 0x00020000);

    // Staging registers.  All loops below have compile-time trip counts and compile-time
    // guards wherever a whole pass over the block fits, so the arrays stay in VGPRs and no
    // load sits behind a divergent branch.
    f32x4 ra[A_PER_THREAD], rb[B_PER_THREAD];
#define B2F_ISSUE_LOADS(c_)                                                                         \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        const int so = (int)(cc * cstr * 4);                                                        \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i)                                    \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? a_rsrc1 : a_rsrc0,     \
                                                                                    (int)(s1 ? a_off1[i] : a_off0[i]), so, 0)); \
        const int wo = c__ * (B_F4 * 16);                                                           \
        _Pragma("unroll") for (int i = 0; i < B_PER_THREAD; ++i) {                                  \
            if ((i + 1) * NTHR <= B_F4)                                                             \
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (tid + i * NTHR) * 16, wo, 0)); \
            else                                                                                    \
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, min(tid + i * NTHR, B_F4 - 1) * 16, wo, 0)); \
        }                                                                                           \
    } while (0)
#define B2F_WRITE_LDS(buf_)                                                                         \
    do {                                                                                            \
        f32x4 *la = lds + (buf_) * BUF_F4;                                                          \
        f32x4 *lb = la + G::A_F4;                                                                   \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i) {                                  \
            const f32x4 v = a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};                            \
            if ((i + 1) * NTHR <= G::A_F4) la[a_lds[i]] = v;                                        \
            else if (a_lds[i] >= 0) la[a_lds[i]] = v;                                               \
        }                                                                                           \
        _Pragma("unroll") for (int i = 0; i < B_PER_THREAD; ++i) {                                  \
            if ((i + 1) * NTHR <= B_F4) lb[tid + i * NTHR] = rb[i];                                 \
            else if (tid + i * NTHR < B_F4) lb[tid + i * NTHR] = rb[i];                             \
        }                                                                                           \
    } while (0)

    // ---- accumulators start at the bias (same order as y = b + sum in nn) ----
    f32x16 acc[NT];
    float bias_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_v[t] = p.bias[nb * NTOT + t * 32 + n];

    // this lane's A pixel (M index m = lane & 31) inside the patch: wave w owns patch rows
    // w (TW == 32) or 2w, 2w+1 (TW == 16)
    const int m_ty = (TW == 32) ? wave : (2 * wave + (n >> 4));
    const int m_tx = (TW == 32) ? n : (n & 15);
    const int a_off = half * NPIX + (m_ty * S) * PW + m_tx * S;
    const int b_off = G::A_F4 + half * NTOT + n;

    // item i = (tile i / nchunks, chunk i % nchunks); `lt`, `lc`: tile / chunk of the item whose loads are issued next
    const int nitems = my_tiles * nchunks;
    int lt = 0, lc = 0;
    auto advance_load = [&]() {
        if (++lc == nchunks) { lc = 0; ++lt; if (lt < my_tiles) enter_tile(tile0 + lt); }
    };
    B2F_ISSUE_LOADS(0);
    B2F_WRITE_LDS(0);
    __syncthreads();
    advance_load();
    if (nitems > 1) B2F_ISSUE_LOADS(lc);
    int c = 0;
    for (int it = 0; it < nitems; ++it) {
        if (c == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bias_v[t];
        }
        const f32x4 *aptr = lds + (it & 1) * BUF_F4 + a_off;
        const f32x4 *bptr = lds + (it & 1) * BUF_F4 + b_off;
        f32x4 a = aptr[0], b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bptr[t * 32];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            f32x4 an = a, bn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = b[t];
            if (tap < 8) {   // operands of the next tap, in flight under this tap's MFMAs
                const int ky = (tap + 1) / 3, kx = (tap + 1) - 3 * ky;
                an = aptr[ky * PW + kx];
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = bptr[(tap + 1) * 2 * NTOT + t * 32];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
            a = an;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = bn[t];
        }
        if (++c == nchunks) {
            c = 0;
            // ---- epilogue of this tile: C/D layout col = lane & 31 (cout), row = (r&3) + 8*(r>>2) + 4*half ----
            float *ob = p.out + (size_t)st_img * p.out_img_stride;
            constexpr int XP = NTOT + 8;                       // floats per pixel row of the exchange (conflict-free both ways)
            constexpr bool XFITS = NW * 32 * XP <= BUF_F4 * 4;
            if (XFITS && vec_ok) {
                // Through LDS to 16-byte stores: a lane holds 16 pixels of ONE channel, so the direct way is 16 NT four-byte
                // stores per lane, 32-byte pieces all over the tile -- measured 4 500 - 9 000 cycles per tile, as long as the
                // tile's MFMAs at stride 2 (in-kernel trace), with the other waves waiting for it at the next barrier.  The
                // LDS buffer this item was read from is free until the end of the next iteration: each wave transposes its
                // 32 pixels x NTOT channels there, then lane (pixel l >> 1, half l & 1) stores the 4 channels of a chunk half:
                // a wave's store instruction covers one 8-channel chunk of its 32 pixels, 1 KB (TW = 32) of contiguous memory.
                __syncthreads();                               // every wave has finished reading this buffer
                float *X = reinterpret_cast<float *>(lds + (it & 1) * BUF_F4) + wave * (32 * XP);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[t][r];
                        if (p.leaky) v = v > 0.f ? v : 0.2f * v;
                        X[m * XP + t * 32 + n] = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes (same-wave LDS traffic is ordered)
                const int m = lane >> 1, hf = lane & 1;
                const int ty = (TW == 32) ? wave : (2 * wave + (m >> 4));
                const int tx = (TW == 32) ? m : (m & 15);
                const int oy = st_oy0 + ty, ox = st_ox0 + tx;
                const bool pix_ok = oy < p.Ho && ox < p.Wo;
                float *opix = ob + (size_t)(oy * p.Wo + ox) * p.out_pix_stride;
#pragma unroll
                for (int k = 0; k < NTOT / 8; ++k) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(X + m * XP + k * 8 + hf * 4);
                    const int co = nb * NTOT + k * 8 + hf * 4;
                    if (pix_ok && co < p.cout) *reinterpret_cast<f32x4 *>(opix + (size_t)(co >> 3) * p.out_chunk_stride + (co & 7)) = v;
                }
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = nb * NTOT + t * 32 + n;
                if (co >= p.cout) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int ty = (TW == 32) ? wave : (2 * wave + (m >> 4));
                    const int tx = (TW == 32) ? m : (m & 15);
                    const int oy = st_oy0 + ty, ox = st_ox0 + tx;
                    if (oy < p.Ho && ox < p.Wo) {
                        float v = acc[t][r];
                        if (p.leaky) v = v > 0.f ? v : 0.2f * v;
                        ob[(size_t)(co >> 3) * p.out_chunk_stride + (size_t)(oy * p.Wo + ox) * p.out_pix_stride + (co & 7)] = v;
                    }
                }
            }
            }
        }
        if (it + 1 < nitems) {
            // buffer (it+1)&1 was last read during item it-1; every wave passed the barrier that
            // closed iteration it-1 after finishing those reads, so it is free to overwrite now
            B2F_WRITE_LDS((it + 1) & 1);
            if (c == 0) { st_img = img; st_ox0 = ox0; st_oy0 = oy0; }   // item it+1 opens the tile the staged loads belong to
            __syncthreads();
            advance_load();
            if (it + 2 < nitems) B2F_ISSUE_LOADS(lc);
        }
    }
#undef B2F_ISSUE_LOADS
#undef B2F_WRITE_LDS
}

template <int S, int NT, int TW, int NW>
static hipError_t launch_t(const ConvLaunch &p, hipStream_t s)
{
    using G = ConvGeom<S, TW, NW>;
    const size_t lds = 2 * sizeof(f32x4) * (G::A_F4 + 9 * 2 * NT * 32);
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];   // one flag per template instance
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_mfma<S, NT, TW, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + G::TH - 1) / G::TH;
    const int ntiles = tiles_x * tiles_y * p.nimg;
    // tiles per block: shallow layers (few K chunks) of launches with many rounds of blocks chain tiles into one pipeline
    ConvLaunch q = p;
    int tpb = 1;
    if (p.tiles_per_block > 0) tpb = p.tiles_per_block;
    else {
        const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
        const long blocks = (long)ntiles * p.nblk;
        if (nchunks <= 8 && blocks >= 8 * 1024) tpb = nchunks <= 2 ? 8 : nchunks <= 4 ? 4 : 2;
    }
    q.tiles_per_block = tpb;
    dim3 grid((unsigned)((ntiles + tpb - 1) / tpb), (unsigned)p.nblk);
    hipLaunchKernelGGL((conv3x3_mfma<S, NT, TW, NW>), grid, dim3(NW * 64), lds, s, q);
    return hipGetLastError();
}

template <int S, int TW, int NW>
static hipError_t launch_nt(const ConvLaunch &p, hipStream_t s)
{
    switch (p.nt) {
        case 1: return launch_t<S, 1, TW, NW>(p, s);
        case 2: return launch_t<S, 2, TW, NW>(p, s);
        case 3: return launch_t<S, 3, TW, NW>(p, s);
        case 4: return launch_t<S, 4, TW, NW>(p, s);
    }
    return hipErrorInvalidValue;
}

// padded output area of a (TH x TW) tiling, used to pick the tile shape
static long padded_area(int Ho, int Wo, int th, int tw)
{
    return (long)((Ho + th - 1) / th * th) * ((Wo + tw - 1) / tw * tw);
}

hipError_t launch_conv3x3(const ConvLaunch &p, hipStream_t s)
{
    if (p.stride != 1 && p.stride != 2) return hipErrorInvalidValue;
    static const int force_nw = getenv("B2F_CONV_NW") ? atoi(getenv("B2F_CONV_NW")) : 0;
    // candidates: 8 waves (8x32 or 16x16 pixels) for big maps, 4 waves (4x32 / 8x16) when the
    // map is small (few blocks) or the 8-wave tiling wastes more than 1/8 of the MFMA work
    const long a8_32 = padded_area(p.Ho, p.Wo, 8, 32), a8_16 = padded_area(p.Ho, p.Wo, 16, 16);
    const long a4_32 = padded_area(p.Ho, p.Wo, 4, 32), a4_16 = padded_area(p.Ho, p.Wo, 8, 16);
    const long best8 = a8_32 <= a8_16 ? a8_32 : a8_16, best4 = a4_32 <= a4_16 ? a4_32 : a4_16;
    const long blocks8 = best8 / 256 * p.nimg * p.nblk;
    bool use8 = blocks8 >= 512 && best8 * 8 <= best4 * 9;
    // stride 2 with few output channels: the (2 TH + 1) x (2 TW + 1) patch of the 8-wave tile leaves one block
    // per CU; two 4-wave blocks hide each other's staging better (measured: 0.79 -> 0.71 ms, 0.56 -> 0.54 ms)
    if (p.stride == 2 && p.nt <= 2) use8 = false;
    if (force_nw == 4) use8 = false;
    if (force_nw == 8) use8 = true;
    if (use8) {
        const bool w32 = a8_32 <= a8_16;
        if (p.stride == 1) return w32 ? launch_nt<1, 32, 8>(p, s) : launch_nt<1, 16, 8>(p, s);
        return w32 ? launch_nt<2, 32, 8>(p, s) : launch_nt<2, 16, 8>(p, s);
    }
    const bool w32 = a4_32 <= a4_16;
    if (p.stride == 1) return w32 ? launch_nt<1, 32, 4>(p, s) : launch_nt<1, 16, 4>(p, s);
    return w32 ? launch_nt<2, 32, 4>(p, s) : launch_nt<2, 16, 4>(p, s);
}

void conv_choose_tiles(int cout, int *nt, int *nblk)
{
    const int n32 = (cout + 31) / 32;
    if (n32 <= 4) { *nt = n32; *nblk = 1; }
    else if (n32 % 3 == 0) { *nt = 3; *nblk = n32 / 3; }
    else { *nt = 4; *nblk = (n32 + 3) / 4; }
}

size_t conv_wpk_floats(int cin_chunks, int nt, int nblk)
{
    return (size_t)nblk * cin_chunks * 9 * 2 * nt * 32 * 4;
}

void conv_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map,
                       int cin_chunks, int nt, int nblk, float *wpk, float *bpk)
{
    const int ntot = nt * 32;
    for (int nb = 0; nb < nblk; ++nb)
        for (int c = 0; c < cin_chunks; ++c)
            for (int tap = 0; tap < 9; ++tap)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < ntot; ++nn)
                        for (int j = 0; j < 4; ++j) {
                            const int co = nb * ntot + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = w[((size_t)co * Ci + ci) * 9 + tap];
                            wpk[(((((size_t)nb * cin_chunks + c) * 9 + tap) * 2 + h) * ntot + nn) * 4 + j] = v;
                        }
    for (int i = 0; i < nblk * ntot; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
