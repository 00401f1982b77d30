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
    int level;    // pyramid level (2..7 for features, 7..3 for decoders)
    int idx;      // conv index inside the unit (1..2 features, 1..6 decoders)
    int ci, co;   // Torch nInputPlane / nOutputPlane
    long long w_off, b_off;   // offsets into the flat canonical buffer
};

std::vector<ConvDesc> weight_layout(bool past_flow, long long *total);
long long param_count(bool past_flow);
void random_weights(unsigned long long seed, bool past_flow, float gain, float *out);

// .t7 reader (b2f_t7.cpp): returns false and fills err on failure.
bool load_t7(const std::string &path, std::vector<float> &flat, bool &past_flow, std::string &err);

}  // namespace b2f
