// conv3x3 as an implicit GEMM on the BF16 matrix pipe with exactly split fp32 operands (round 4): the direct layers that the fp32
// MFMA bounds -- the stride-2 first convs of the convUnits, nn.SpatialConvolution(Ci,Co,3,3,2,2,1,1) + LeakyReLU(0.2) of
// /root/reference/models/pwc.lua:60 (32 -> 64, 64 -> 96, 96 -> 128, 128 -> 192) -- and any stride-1 layer routed here.
// conv3x3_mfma (b2f_conv.hip) runs them at 48-74 % of the fp32 pipe; the same products cost 3 v_mfma_f32_32x32x16_bf16 (96 cycles)
// instead of 4 v_mfma_f32_32x32x2_f32 (256) per (32 outputs, 32 pixels, 8 channels, tap) when every fp32 operand is split into three
// bf16 terms and six of the nine term products are kept (fp32-level accuracy: b2f_wino4s.hip, b2f_conv16b.hip).
//
// GEMM view: D[co 32][pixel 32] += W[co][k 16] X[k 16][pixel]; lane (row / column = lane & 31, k4 = lane >> 5) holds channels 4 k4 ..
// 4 k4 + 3 of the 8-channel chunk as windows of bf16 pairs  Wa = [m01 m23 h01 h23], Wb = [h01 h23 l01 l23]  (Xa, Xb alike):
// Wa Xa + Wb Xa + Wa Xb.  Weights as the A operand: a lane then ends up with four runs of four consecutive output channels of ONE
// pixel -- 16-byte stores straight from the accumulators, no transpose through LDS.
//   * block = 4 waves = 8 output rows x 32 columns x 64 outputs (NT = 2; NT = 1: 32); wave w owns rows 2w, 2w + 1 (two pixel tiles) and
//     all NT output tiles: 2 NT accumulators, every weight window feeds two pixel tiles, every pixel window NT output tiles;
//   * K is walked in chunks of 8 channels.  Per chunk the input patch with halo ((8 - 1) S + 3 rows x (32 - 1) S + 3 columns) is
//     brought global -> registers (requested under the MFMAs of the previous chunk) -> split (22 VALU per channel quad) -> LDS,
//     two planes [window a | b][k4][pixel] of 16 bytes; stride 2 stores even | odd columns apart so that the taps read consecutive
//     slots.  One buffer (70.7 KB at stride 2), two blocks per CU: one block's staging runs under the other's MFMAs (the bf16
//     MFMA leaves the VALU free);
//   * the weight windows come pre-split from the host packing [n-block][chunk][tap][k4][co 64] x 32 bytes straight from L2 into a
//     register ring two taps ahead (43 B/clk per CU at full pipe).
#include "b2f_internal.h"

#include <cstring>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int S>
struct ConvbGeom {
    static constexpr int TH = 8, TW = 32;
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;   // 10 x 34 | 17 x 65
    static constexpr int NPIX = PH * PW;                                // 340 | 1105
    static constexpr int PLANE = NPIX + ((4 - NPIX % 16) + 16) % 16;    // = 4 (mod 16): the k4 planes start 16 banks apart
    static constexpr int LDS_BYTES = 2 * 2 * PLANE * 16;                // 21 824 | 70 912
    static constexpr int NITEM = 2 * NPIX;                              // (pixel, k4) staging items per chunk
    static constexpr int NJ = (NITEM + 255) / 256;                      // 3 | 9
    static constexpr int EVEN = (PW + 1) / 2;                           // stride 2: even columns first
    __device__ __host__ static constexpr int colslot(int px) { return S == 1 ? px : (px >> 1) + (px & 1) * EVEN; }
};

__device__ __forceinline__ unsigned convb_pk(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void convb_split(const f32x4 v, u32x4 &wa, u32x4 &wb)
{
    const unsigned h01 = convb_pk(v[0], v[1]), h23 = convb_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = convb_pk(r0, r1), m23 = convb_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    wa = u32x4{m01, m23, h01, h23};
    wb = u32x4{h01, h23, convb_pk(l0, l1), convb_pk(l2, l3)};
}

#ifndef B2F_CONVB_ABLATE
#define B2F_CONVB_ABLATE 0   // profiling only (wrong results): 1 no patch loads, 2 no weight loads, 4 no MFMAs, 8 no split / LDS writes
#endif
#define CONVB_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)

template <int S, int NT, bool DB>
__global__ __launch_bounds__(256, 2) void conv3x3_bf6(const ConvLaunch p)
{
    using G = ConvbGeom<S>;
    constexpr int PW = G::PW, PLANE = G::PLANE, NJ = G::NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *L = reinterpret_cast<u32x4 *>(smem);                    // [buffer 1 | 2][window 2][k4 2][PLANE]
    constexpr int BUF = 4 * PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, k4 = lane >> 5;

    const int tiles_x = (p.Wo + G::TW - 1) / G::TW, tiles_y = (p.Ho + G::TH - 1) / G::TH;
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int nb = blockIdx.y + p.nb0;                             // block of 64 outputs (NT = 1 computes its first 32)
    const int ox0 = tx_i * G::TW, oy0 = ty_i * G::TH;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);

    // ---- staging items of this thread: idx = tid + 256 j = (pixel idx >> 1, k4 = idx & 1) ----
    unsigned s_off0[NJ], s_off1[NJ];
    int s_dst[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = tid + 256 * j;
        const int pix = min(idx, G::NITEM - 1) >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = oy0 * S - 1 + py, gx = ox0 * S - 1 + px;
        const bool ok = idx < G::NITEM && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned gp = (unsigned)(gy * p.W + gx);
        s_off0[j] = ok ? (gp * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u : 0xfffffff0u;   // past the resource: reads as zero
        s_off1[j] = ok ? (gp * (unsigned)p.seg[1].pix_stride + (tid & 1) * 4) * 4u : 0xfffffff0u;
        s_dst[j] = idx < G::NITEM ? (tid & 1) * PLANE + py * PW + G::colslot(px) : -1;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t a_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
    f32x4 raw[NJ];
    auto request = [&](const int c) {
        const bool s1 = c >= p.seg[0].nchunks;
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;
        const int so = (int)((s1 ? c - p.seg[0].nchunks : c) * cstr * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (B2F_CONVB_ABLATE & 1) raw[j] = f32x4{1.f, 2.f, 3.f, 4.f};
            else raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? a_rsrc1 : a_rsrc0, (int)(s1 ? s_off1[j] : s_off0[j]), so, 0));
        }
    };

    // ---- weights: [n-block][chunk][tap 9][k4 2][co 64] x (Wa | Wb); lane (co = 32 t + n, k4) ----
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wpk_bf6) + (size_t)nb * nchunks * (9 * 2 * 64 * 32)), 0, 0x7fffffff, 0x00020000);
    const int w_lane = (k4 * 64 + n) * 32;
    u32x4 wa[3][NT], wb[3][NT];                                    // ring: taps t, t + 1, t + 2
    auto load_w = [&](const int slot, const int c, const int tap) {
        const int so = (c * 9 + tap) * (2 * 64 * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (B2F_CONVB_ABLATE & 2) { wa[slot][t] = u32x4{1u, 2u, 3u, (unsigned)tap}; wb[slot][t] = u32x4{4u, 5u, 6u, (unsigned)c}; continue; }
            wa[slot][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane + t * 32 * 32, so, 0));
            wb[slot][t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane + t * 32 * 32 + 16, so, 0));
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][t][r] = 0.f;

    // window a of (patch row (2 wave + pt) S, patch column n S): + ky PW rows, + colslot(kx) columns
    const u32x4 *x_base = L + k4 * PLANE + (2 * wave * S) * PW + (S == 1 ? n : n);

    auto split_to_lds = [&](u32x4 *dst) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!(B2F_CONVB_ABLATE & 8) && ((j + 1) * 256 <= G::NITEM || s_dst[j] >= 0)) {
                u32x4 a, b;
                convb_split(raw[j], a, b);
                dst[s_dst[j]] = a;
                dst[2 * PLANE + s_dst[j]] = b;
            }
        }
    };
    auto mfmas = [&](const int slot, const u32x4 (&xa)[2], const u32x4 (&xb)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (B2F_CONVB_ABLATE & 4) { acc[pt][t][0] += __builtin_bit_cast(float, wa[slot][t][0] ^ xa[pt][1] ^ wb[slot][t][2] ^ xb[pt][3]); continue; }
                CONVB_MF(acc[pt][t], wa[slot][t], xa[pt]);
                CONVB_MF(acc[pt][t], wb[slot][t], xa[pt]);
                CONVB_MF(acc[pt][t], wa[slot][t], xb[pt]);
            }
    };
    request(0);
    load_w(0, 0, 0);
    load_w(1, 0, 1);
    if (DB) {
        // ---- double-buffered patch, ONE barrier per chunk: chunk c + 1 is split into the other buffer between the taps of chunk c
        // (its loads were requested a chunk earlier), the pixel windows of tap t + 1 are read before the MFMAs of tap t ----
        split_to_lds(L);
        if (nchunks > 1) request(1);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const u32x4 *xb0 = x_base + (c & 1) * BUF;
            u32x4 xa[2][2], xb[2][2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const u32x4 *xp = xb0 + (pt * S) * PW;
                xa[0][pt] = xp[0];
                xb[0][pt] = xp[2 * PLANE];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                {
                    const int tn = tap + 2;
                    if (tn < 9) load_w(tn % 3, c, tn);
                    else if (c + 1 < nchunks) load_w(tn % 3, c + 1, tn - 9);
                }
                if (tap < 8) {
                    const int ky = (tap + 1) / 3, kx = (tap + 1) - 3 * ky;
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) {
                        const u32x4 *xp = xb0 + (pt * S + ky) * PW + G::colslot(kx);
                        xa[(tap + 1) & 1][pt] = xp[0];
                        xb[(tap + 1) & 1][pt] = xp[2 * PLANE];
                    }
                }
                mfmas(tap % 3, xa[tap & 1], xb[tap & 1]);
                if (tap == 4 && c + 1 < nchunks) split_to_lds(L + ((c + 1) & 1) * BUF);
                if (tap == 5 && c + 2 < nchunks) request(c + 2);
            }
            __syncthreads();               // buffer (c + 1) & 1 is complete; everyone is done reading buffer c & 1
        }
    } else {
    for (int c = 0; c < nchunks; ++c) {
        // ---- split the staged chunk into LDS ----
        split_to_lds(L);
        __syncthreads();
        if (c + 1 < nchunks) request(c + 1);                        // lands under the MFMAs below
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // weights two taps ahead (the ring's third slot); the last two taps of a chunk fetch the first two of the next
            {
                const int tn = tap + 2;
                if (tn < 9) load_w(tn % 3, c, tn);
                else if (c + 1 < nchunks) load_w(tn % 3, c + 1, tn - 9);
            }
            const int ky = tap / 3, kx = tap - 3 * ky;
            u32x4 xa[2], xb[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const u32x4 *xp = x_base + (pt * S + ky) * PW + G::colslot(kx);
                xa[pt] = xp[0];
                xb[pt] = xp[2 * PLANE];
            }
            mfmas(tap % 3, xa, xb);
        }
        __syncthreads();                                           // everyone is done reading the patch
    }
    }

    // ---- epilogue: lane (pixel n, k4) holds rows (r & 3) + 8 (r >> 2) + 4 k4 of every 32-output tile: chunk j, half k4 ----
    float *ob = p.out + (size_t)img * p.out_img_stride;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int oy = oy0 + 2 * wave + pt, ox = ox0 + n;
        if (oy < p.Ho && ox < p.Wo) {
            float *opix = ob + (size_t)(oy * p.Wo + ox) * p.out_pix_stride;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = nb * 64 + t * 32 + 8 * j + 4 * k4;
                    if (co < p.cout) {
                        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias_bf6 + co);
                        f32x4 v = f32x4{acc[pt][t][4 * j], acc[pt][t][4 * j + 1], acc[pt][t][4 * j + 2], acc[pt][t][4 * j + 3]} + bias;
                        if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);
                        *reinterpret_cast<f32x4 *>(opix + (size_t)(co >> 3) * p.out_chunk_stride + (co & 7)) = v;
                    }
                }
        }
    }
}

bool convb_supported(const ConvLaunch &p)
{
    if ((p.stride != 1 && p.stride != 2) || !p.wpk_bf6 || !p.bias_bf6) return false;
    if (p.Ho != (p.H - 1) / p.stride + 1 || p.Wo != (p.W - 1) / p.stride + 1) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride && false) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0) return false;   // 16-byte stores
    for (int i = 0; i < p.nseg; ++i)
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 >= 2147483648.0 ||
            (double)p.H * p.W * p.seg[i].pix_stride * 4.0 >= 2147483648.0) return false;
    return true;
}

template <int S, int NT, bool DB>
static hipError_t convb_launch_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using G = ConvbGeom<S>;
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_bf6<S, NT, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, (DB ? 2 : 1) * G::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.Wo + G::TW - 1) / G::TW) * ((p.Ho + G::TH - 1) / G::TH);
    ConvLaunch q = p;
    q.nb0 = nb0;
    hipLaunchKernelGGL((conv3x3_bf6<S, NT, DB>), dim3((unsigned)(tiles * p.nimg), (unsigned)nblk), dim3(256), (DB ? 2 : 1) * G::LDS_BYTES, s, q);
    return hipGetLastError();
}

hipError_t launch_conv3x3_bf6(const ConvLaunch &p, hipStream_t s)
{
    if (!convb_supported(p)) return hipErrorInvalidValue;
    // whole blocks of 64 outputs on the two-tile kernel; a remainder of at most 32 outputs on the one-tile kernel (first half of its block)
    const int rem = p.cout % 64, full = p.cout / 64 + (rem > 32 ? 1 : 0);
    hipError_t e = hipSuccess;
    if (full > 0) e = p.stride == 1 ? convb_launch_t<1, 2, true>(p, 0, full, s) : convb_launch_t<2, 2, false>(p, 0, full, s);
    if (e == hipSuccess && rem > 0 && rem <= 32) e = p.stride == 1 ? convb_launch_t<1, 1, true>(p, full, 1, s) : convb_launch_t<2, 1, false>(p, full, 1, s);
    return e;
}

int convb_nblk(int cout) { return (cout + 63) / 64; }
size_t convb_wpk_floats(int cin_chunks, int cout) { return (size_t)convb_nblk(cout) * cin_chunks * (9 * 2 * 64 * 32 / 4); }

static inline unsigned short convb_bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float convb_bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// [n-block of 64][chunk][tap 9][k4 2][co 64] x { Wa = (m01 m23 h01 h23), Wb = (h01 h23 l01 l23) } of bf16 pairs, w = h + m + l exactly;
// bpk: bias padded to whole blocks
void convb_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk)
{
    const int nblk = convb_nblk(Co);
    unsigned short *out = reinterpret_cast<unsigned short *>(wpk);
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c)
            for (int tap = 0; tap < 9; ++tap)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < 64; ++nn) {
                        unsigned short *q = out + ((((size_t)(nbk * cin_chunks + c) * 9 + tap) * 2 + h) * 64 + nn) * 16;
                        for (int j = 0; j < 4; ++j) {
                            const int co = nbk * 64 + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = w[((size_t)co * Ci + ci) * 9 + tap];
                            const unsigned short hh = convb_bf16_rne(v);
                            const float r1 = v - convb_bf16_f32(hh);
                            const unsigned short mm = convb_bf16_rne(r1);
                            const float r2 = r1 - convb_bf16_f32(mm);
                            const unsigned short ll = convb_bf16_rne(r2);
                            q[j] = mm; q[4 + j] = hh;              // Wa
                            q[8 + j] = hh; q[12 + j] = ll;         // Wb
                        }
                    }
    for (int i = 0; i < nblk * 64; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
