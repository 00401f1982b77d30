// Host-only helpers shared by the C-ABI layer: canonical weight layout, deterministic
// init, the .t7 reader.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace b2f {

constexpr int kFeatH[8] = {0, 3, 16, 32, 64, 96, 128, 192};   // featMaps, pwc.lua:29,89
constexpr int kDecH[7] = {0, 128, 128, 96, 64, 32, 2};        // decoder(), pwc.lua:76-85
constexpr int kNDh = 162;

enum { KIND_FEAT = 0, KIND_OCC = 1, KIND_FLOW = 2, KIND_PAST = 3 };

struct ConvDesc {
    int kind;     // KIND_*
    int level;    // pyramid level (2..7 for features -- 1..7 with pwc_skip = 0 --, levels..l_st for decoders)
    int idx;      // conv index inside the unit (1..2 features, 1..6 decoders)
    int ci, co;   // Torch nInputPlane / nOutputPlane
    long long w_off, b_off;   // offsets into the flat canonical buffer
};

// The option table of createModelMulti (models/pwc.lua:88-121) that shapes the graph; frames = 3 is fixed.
// Defaults = the shipped models (opts.lua:83-98).
struct GraphOpts {
    int win = 9, levels = 7, skip = 2;
    int two_frame = 0, sum_cvs = 0, residual = 0, occ_input = 0, rescale_flow = 0;
    int siamese = 1;                                                               // pwc_siamese, pwc.lua:99,115
    bool past_flow = false;
    float flownet_factor = 20.f;
    int l_st() const { return skip + 1; }                                          // pwc.lua:136 (skip >= 0)
    // featMaps[l], pwc.lua:89,120-127: pwc_skip = 0 gives the level-1 unit featMaps[2] maps, pwc_siamese = 0 makes every level the image
    int feat(int l) const { return !siamese ? 3 : (l == 1 && skip == 0) ? kFeatH[2] : kFeatH[l]; }
    int feat_first() const { return skip == 0 ? 1 : 2; }                           // first level with a convUnit (pwc.lua:171-183)
    int nd() const { return win * win; }
    int nd_flow() const { return (two_frame || sum_cvs) ? nd() : 2 * nd(); }       // pwc.lua:254-285
    int nd_occ() const { return two_frame ? nd() : 2 * nd(); }
    int occ_in(int l) const                                                        // pwc.lua:288-305
    {
        int n = nd_occ() + feat(l) + (two_frame ? feat(l) : 0);
        if (l != levels) n += 2 + (occ_input ? 2 : 0);
        return n;
    }
    int flow_in(int l) const { return l == levels ? nd_flow() : nd_flow() + feat(l) + 2; }   // pwc.lua:325-337
    int n_outputs() const { return (levels - l_st() + 1) * (past_flow ? 5 : 4); }              // pwc.lua:459-489
    // the graph the fused fast path of b2f_api.hip is written for (any past_flow)
    bool shipped() const
    {
        return win == 9 && levels == 7 && skip == 2 && !two_frame && !sum_cvs && !residual && !occ_input && !rescale_flow &&
               siamese && flownet_factor == 20.f;
    }
    bool valid() const { return win >= 1 && (win & 1) && win <= 15 && levels >= 2 && levels <= 7 && skip >= 0 && skip + 1 <= levels; }
};
// "win=5,levels=4,skip=2,two_frame=0,sum_cvs=1,residual=1,occ_input=1,rescale_flow=1,flownet_factor=20" (any subset;
// also siamese=0|1 and the reference's option names pwc_ws, pwc_skip, pwc_sum_cvs, pwc_siamese).  past_flow is not set here (it comes with the weights).
bool parse_graph_opts(const char *text, GraphOpts &o, std::string &err);
std::string graph_opts_string(const GraphOpts &o);

std::vector<ConvDesc> weight_layout(const GraphOpts &o, long long *total);
long long param_count(const GraphOpts &o);
void random_weights(unsigned long long seed, const GraphOpts &o, float gain, float *out);
std::vector<ConvDesc> weight_layout(bool past_flow, long long *total);
long long param_count(bool past_flow);
void random_weights(unsigned long long seed, bool past_flow, float gain, float *out);

// .t7 reader (b2f_t7.cpp): returns false and fills err on failure.
bool load_t7(const std::string &path, std::vector<float> &flat, bool &past_flow, std::string &err);
// any graph shape: infer = true takes win / levels / skip from the file, false checks the file against g (see b2f_t7.cpp)
bool load_t7_ex(const std::string &path, GraphOpts &g, bool infer, std::vector<float> &flat, std::string &err);

}  // namespace b2f
