// Host-side pieces of the computeFlow boundary: the deterministic weight initialiser and
// the canonical weight layout.  No GPU code here (the pre/post-processing that back2future.lua
// does around model:forward runs on the device, b2f_boundary.hip).
#include "b2f_host.h"

#include <cmath>
#include <cstring>

namespace b2f {

// ---- canonical flat layout: mirrors how models/pwc.lua builds the graph ----
static int occ_in_ch(int l) { return kNDh + kFeatH[l] + (l != 7 ? 2 : 0); }     // pwc.lua:288-304
static int flow_in_ch(int l) { return l == 7 ? kNDh : kNDh + kFeatH[l] + 2; }   // pwc.lua:325-337

std::vector<ConvDesc> weight_layout(bool past_flow, long long *total)
{
    std::vector<ConvDesc> v;
    long long off = 0;
    auto add = [&](int kind, int level, int idx, int ci, int co) {
        ConvDesc d;
        d.kind = kind; d.level = level; d.idx = idx; d.ci = ci; d.co = co;
        d.w_off = off; off += (long long)co * ci * 9;
        d.b_off = off; off += co;
        v.push_back(d);
    };
    for (int l = 2; l <= 7; ++l) {   // convUnit, pwc.lua:58-65
        add(KIND_FEAT, l, 1, kFeatH[l - 1], kFeatH[l]);
        add(KIND_FEAT, l, 2, kFeatH[l], kFeatH[l]);
    }
    for (int l = 7; l >= 3; --l) {   // decoder(), pwc.lua:76-85
        const int kinds[3] = {KIND_OCC, KIND_FLOW, KIND_PAST};
        const int nin[3] = {occ_in_ch(l), flow_in_ch(l), flow_in_ch(l)};
        for (int k = 0; k < (past_flow ? 3 : 2); ++k) {
            int ci = nin[k];
            for (int i = 1; i <= 6; ++i) { add(kinds[k], l, i, ci, kDecH[i]); ci = kDecH[i]; }
        }
    }
    if (total) *total = off;
    return v;
}

long long param_count(bool past_flow)
{
    long long t = 0;
    weight_layout(past_flow, &t);
    return t;
}

// ---- splitmix64 counter generator; must match back2future_amd/weights.py ----
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void random_weights(unsigned long long seed, bool past_flow, float gain, float *out)
{
    long long total = 0;
    const std::vector<ConvDesc> lay = weight_layout(past_flow, &total);
    const uint64_t base = (uint64_t)seed * 0x100000001B3ull;
    for (const ConvDesc &d : lay) {
        // nn.SpatialConvolution:reset() [3P]: stdv = 1/sqrt(kW*kH*nInputPlane), U(-stdv, stdv)
        const float s = gain / sqrtf((float)(d.ci * 9));
        const long long n = (long long)d.co * d.ci * 9 + d.co;
        for (long long i = 0; i < n; ++i) {
            const uint64_t z = splitmix64(base + (uint64_t)(d.w_off + i));
            const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
            const float t = 2.0f * u - 1.0f;
            out[d.w_off + i] = t * s;
        }
    }
}

}  // namespace b2f
