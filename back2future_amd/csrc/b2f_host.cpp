// Host-side pieces of the computeFlow boundary: the deterministic weight initialiser and
// the canonical weight layout.  No GPU code here (the pre/post-processing that back2future.lua
// does around model:forward runs on the device, b2f_boundary.hip).
#include "b2f_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace b2f {

// ---- canonical flat layout: mirrors how models/pwc.lua builds the graph ----
std::vector<ConvDesc> weight_layout(const GraphOpts &o, long long *total)
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
    for (int l = o.feat_first(); o.siamese && l <= o.levels; ++l) {   // convUnit, pwc.lua:58-65,169-183
        add(KIND_FEAT, l, 1, l == 1 ? 3 : o.feat(l - 1), o.feat(l));
        add(KIND_FEAT, l, 2, o.feat(l), o.feat(l));
    }
    for (int l = o.levels; l >= o.l_st(); --l) {   // decoder(), pwc.lua:76-85
        const int kinds[3] = {KIND_OCC, KIND_FLOW, KIND_PAST};
        const int nin[3] = {o.occ_in(l), o.flow_in(l), o.flow_in(l)};
        for (int k = 0; k < (o.past_flow ? 3 : 2); ++k) {
            int ci = nin[k];
            for (int i = 1; i <= 6; ++i) { add(kinds[k], l, i, ci, kDecH[i]); ci = kDecH[i]; }
        }
    }
    if (total) *total = off;
    return v;
}

std::vector<ConvDesc> weight_layout(bool past_flow, long long *total)
{
    GraphOpts o;
    o.past_flow = past_flow;
    return weight_layout(o, total);
}

long long param_count(const GraphOpts &o)
{
    long long t = 0;
    weight_layout(o, &t);
    return t;
}

long long param_count(bool past_flow)
{
    long long t = 0;
    weight_layout(past_flow, &t);
    return t;
}

bool parse_graph_opts(const char *text, GraphOpts &o, std::string &err)
{
    if (!text) return true;
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t q = s.find(',', pos);
        if (q == std::string::npos) q = s.size();
        const std::string item = s.substr(pos, q - pos);
        pos = q + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) { err = "graph option '" + item + "' has no value"; return false; }
        const std::string k = item.substr(0, eq), val = item.substr(eq + 1);
        char *end = nullptr;
        const double d = strtod(val.c_str(), &end);
        double num = d;
        if (val == "true") num = 1;
        else if (val == "false") num = 0;
        else if (end == val.c_str() || *end) { err = "graph option '" + item + "': not a number"; return false; }
        const int iv = (int)num;
        if (k == "win" || k == "pwc_ws") o.win = iv;
        else if (k == "levels") o.levels = iv;
        else if (k == "skip" || k == "pwc_skip") o.skip = iv;
        else if (k == "two_frame") o.two_frame = iv != 0;
        else if (k == "sum_cvs" || k == "pwc_sum_cvs") o.sum_cvs = iv != 0;
        else if (k == "residual") o.residual = iv != 0;
        else if (k == "occ_input") o.occ_input = iv != 0;
        else if (k == "rescale_flow") o.rescale_flow = iv != 0;
        else if (k == "siamese" || k == "pwc_siamese") o.siamese = iv != 0;
        else if (k == "flownet_factor") o.flownet_factor = (float)num;
        else { err = "unknown graph option '" + k + "'"; return false; }
    }
    if (!o.valid()) {
        err = "unsupported graph options (need odd win <= 15, 2 <= levels <= 7, 0 <= skip < levels; frames = 3 is fixed)";
        return false;
    }
    return true;
}

std::string graph_opts_string(const GraphOpts &o)
{
    char buf[256];
    snprintf(buf, sizeof buf, "win=%d,levels=%d,skip=%d,two_frame=%d,sum_cvs=%d,residual=%d,occ_input=%d,rescale_flow=%d,siamese=%d,flownet_factor=%g,past_flow=%d",
             o.win, o.levels, o.skip, o.two_frame, o.sum_cvs, o.residual, o.occ_input, o.rescale_flow, o.siamese, (double)o.flownet_factor, o.past_flow ? 1 : 0);
    return buf;
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
    GraphOpts o;
    o.past_flow = past_flow;
    random_weights(seed, o, gain, out);
}

void random_weights(unsigned long long seed, const GraphOpts &o, float gain, float *out)
{
    long long total = 0;
    const std::vector<ConvDesc> lay = weight_layout(o, &total);
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
