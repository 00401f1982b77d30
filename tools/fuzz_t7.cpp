// ASan/UBSan fuzz of the .t7 reader (host code only; sanitizers run on the CPU build):
//   python -c "import sys; sys.path.insert(0, \".\"); from tests import t7_writer; from back2future_amd import weights as W; t7_writer.save(\"/tmp/good.t7\", W.random_init(3, True, 1.0), True, dpt=True)"
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iback2future_amd/csrc tools/fuzz_t7.cpp back2future_amd/csrc/b2f_t7.cpp back2future_amd/csrc/b2f_host.cpp -o /tmp/fuzz_t7 && /tmp/fuzz_t7 /tmp/good.t7 300
#include "b2f_host.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
int main(int argc, char **argv)
{
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<float> flat; bool past; std::string err;
    if (!b2f::load_t7(argv[1], flat, past, err)) { printf("valid file rejected: %s\n", err.c_str()); return 1; }
    printf("valid: %zu floats past=%d\n", flat.size(), (int)past);
    unsigned s = 12345; int ok = 0, rej = 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    for (int it = 0; it < iters; ++it) {
        std::vector<char> b = good;
        s = s * 1664525u + 1013904223u;
        const int mode = (s >> 8) % 3;
        if (mode == 0) b.resize((size_t)((s >> 10) % b.size()));
        else { const int nf = 1 + (s >> 12) % 8; for (int k = 0; k < nf; ++k) { s = s * 1664525u + 1013904223u; b[(size_t)(s >> 4) % std::min<size_t>(b.size(), mode == 1 ? 4096 : b.size())] ^= (char)(1 << ((s >> 28) & 7)); } }
        FILE *o = fopen("/tmp/fuzz.t7", "wb"); fwrite(b.data(), 1, b.size(), o); fclose(o);
        std::vector<float> fl; bool p; std::string e;
        if (b2f::load_t7("/tmp/fuzz.t7", fl, p, e)) ++ok; else ++rej;
    }
    printf("corrupted variants: %d accepted, %d rejected, no crash\n", ok, rej);
    return 0;
}
