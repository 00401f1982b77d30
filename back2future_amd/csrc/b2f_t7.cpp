// Torch7 ".t7" reader for the shipped multi-frame PWC models: replaces `torch.load(modelPath)`
// + `model:get(1)` of /root/reference/back2future.lua:113-116 without Lua/Torch.
//
// [3P] The binary serialization format lives in torch7 (File.lua, not in the reference tree);
// restated from its published behaviour: little endian; every object starts with an int32 type
// tag {0 nil, 1 number (double), 2 string (int32 len + bytes), 3 table, 4 torch object,
// 5 boolean (int32), 6/7/8 function}; tables, torch objects and functions carry an int32
// reference index (an index seen before = the same object again -- this is how the siamese
// clones of pwc.lua:187-195 share one storage); a torch object is a version string "V 1", a class
// name, then either a class-specific payload (Tensor: int32 nDim, int64 size[], int64 stride[],
// int64 1-based storageOffset, storage object; Storage: int64 n + raw elements; Cuda* classes
// use the float/.. payload of their CPU twins) or, for every nn.* / cudnn.* / nngraph.* /
// graph.* object, a table of its fields.
// No .t7 file exists in the reference tree (the three models are Dropbox links, README.md:49-52),
// so this reader is validated against a writer that produces the same structures
// (tests/t7_writer.py), not against a real file: see DESIGN.md.
//
// Role recovery: the gModule's forward nodes are walked; nn.Sequential
// modules with 2 convolutions are the convUnits (level from nInputPlane -> nOutputPlane), with 6
// convolutions decoders (level from the first nInputPlane: 354/162 -> 7, 292, 260, 228, 196 -> 6..3).
// Decoders of one level have identical shapes, so the role comes from the graph: consumer is a
// SpatialSoftMax -> occlusion decoder; otherwise the MulConstant nodes reachable through the
// up-sampling chain decide: a positive constant (20*(3-2)/2^k, pwc.lua:404,443) -> future flow,
// only negative ones -> past flow.
#include "b2f_host.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <set>

namespace b2f {
namespace {

struct Obj;
typedef std::shared_ptr<Obj> Ref;

struct Obj {
    enum Kind { NIL, NUM, STR, BOOL, TABLE, TORCH, TENSOR, STORAGE, FUNC } kind = NIL;
    double num = 0;
    bool b = false;
    std::string str;                          // STR value / class name of TORCH, TENSOR, STORAGE
    std::vector<std::pair<Ref, Ref>> items;   // TABLE
    Ref payload;                              // TORCH: its field table (may be null)
    // TENSOR
    std::vector<long long> size, stride;
    long long offset = 0;                     // 0-based
    Ref storage;
    // STORAGE
    std::vector<float> data;
};

struct Reader {
    FILE *f = nullptr;
    std::string err;
    std::map<int, Ref> seen;
    std::vector<Ref> all;   // every object created: the nngraph node tables reference each other in cycles
    bool ok = true;

    ~Reader()
    {
        // break the reference cycles, or the shared_ptr graph (~200 MB for a shipped model) is never freed
        for (Ref &o : all) { o->items.clear(); o->payload.reset(); o->storage.reset(); }
    }

    bool rd(void *p, size_t n)
    {
        if (!ok) return false;
        if (fread(p, 1, n, f) != n) { ok = false; err = "unexpected end of file"; return false; }
        return true;
    }
    int i32() { int32_t v = 0; rd(&v, 4); return v; }
    long long i64() { int64_t v = 0; rd(&v, 8); return v; }
    double f64() { double v = 0; rd(&v, 8); return v; }
    std::string str()
    {
        const int n = i32();
        if (!ok || n < 0 || n > (1 << 28)) { ok = false; if (err.empty()) err = "bad string length"; return ""; }
        std::string s((size_t)n, '\0');
        if (n) rd(&s[0], (size_t)n);
        return s;
    }

    static int elem_size(const std::string &cls)
    {
        if (cls.find("Double") != std::string::npos) return 8;
        if (cls.find("Long") != std::string::npos) return 8;
        if (cls.find("Half") != std::string::npos) return 2;
        if (cls.find("Short") != std::string::npos) return 2;
        if (cls.find("Byte") != std::string::npos || cls.find("Char") != std::string::npos) return 1;
        return 4;   // Float, Cuda (= float), Int
    }

    Ref object(int depth = 0)
    {
        Ref o = std::make_shared<Obj>();
        all.push_back(o);
        if (!ok) return o;
        if (depth > 4000) { ok = false; err = "object nesting too deep"; return o; }
        const int tag = i32();
        switch (tag) {
            case 0: o->kind = Obj::NIL; return o;
            case 1: o->kind = Obj::NUM; o->num = f64(); return o;
            case 2: o->kind = Obj::STR; o->str = str(); return o;
            case 5: o->kind = Obj::BOOL; o->b = i32() != 0; return o;
            case 6:
                // TYPE_FUNCTION (legacy, torch7 File.lua [3P]): no reference index -- int32 size, dumped chunk, then the
                // upvalues object.  Not needed, but must be consumed with the right framing.
                o->kind = Obj::FUNC;
                (void)str();
                (void)object(depth + 1);
                return o;
            case 3: case 4: case 7: case 8: break;
            default: ok = false; err = "unknown type tag " + std::to_string(tag); return o;
        }
        const int idx = i32();
        auto it = seen.find(idx);
        if (it != seen.end()) return it->second;
        seen[idx] = o;
        if (tag == 3) {
            o->kind = Obj::TABLE;
            const int n = i32();
            if (n < 0) { ok = false; err = "negative table size"; return o; }
            for (int i = 0; i < n && ok; ++i) {
                Ref k = object(depth + 1);
                Ref v = object(depth + 1);
                o->items.emplace_back(k, v);
            }
            return o;
        }
        if (tag == 4) {
            std::string version = str();
            std::string cls;
            if (version.compare(0, 2, "V ") == 0) cls = str();
            else cls = version;   // legacy files have no version string
            o->str = cls;
            const bool is_tensor = cls.size() > 6 && cls.compare(0, 6, "torch.") == 0 && cls.find("Tensor") != std::string::npos;
            const bool is_storage = cls.size() > 6 && cls.compare(0, 6, "torch.") == 0 && cls.find("Storage") != std::string::npos;
            if (is_tensor) {
                o->kind = Obj::TENSOR;
                const int nd = i32();
                if (nd < 0 || nd > 16) { ok = false; err = "bad tensor rank"; return o; }
                for (int i = 0; i < nd; ++i) o->size.push_back(i64());
                for (int i = 0; i < nd; ++i) o->stride.push_back(i64());
                o->offset = i64() - 1;
                o->storage = object(depth + 1);
                return o;
            }
            if (is_storage) {
                o->kind = Obj::STORAGE;
                const long long n = i64();
                const int es = elem_size(cls);
                if (n < 0 || n > (1ll << 33)) { ok = false; err = "bad storage size"; return o; }
                std::vector<char> raw((size_t)n * es);
                if (n) rd(raw.data(), raw.size());
                o->data.resize((size_t)n);
                for (long long i = 0; i < n && ok; ++i) {
                    const char *p = raw.data() + (size_t)i * es;
                    if (cls.find("Double") != std::string::npos) { double v; memcpy(&v, p, 8); o->data[i] = (float)v; }
                    else if (cls.find("Long") != std::string::npos) { int64_t v; memcpy(&v, p, 8); o->data[i] = (float)v; }
                    else if (cls.find("Int") != std::string::npos) { int32_t v; memcpy(&v, p, 4); o->data[i] = (float)v; }
                    else if (es == 4) { float v; memcpy(&v, p, 4); o->data[i] = v; }
                    else if (es == 1) o->data[i] = (float)(unsigned char)p[0];
                    else o->data[i] = 0.f;   // half / short: not used by the shipped models
                }
                return o;
            }
            o->kind = Obj::TORCH;
            o->payload = object(depth + 1);   // the table of fields
            return o;
        }
        // TYPE_RECUR_FUNCTION (8) / LEGACY_RECUR_FUNCTION (7): indexed like tables; dumped chunk + upvalue table
        o->kind = Obj::FUNC;
        (void)str();
        (void)object(depth + 1);
        return o;
    }
};

// ---- helpers over the object graph ----
Ref field(const Ref &o, const char *name)
{
    if (!o) return nullptr;
    const Ref &t = (o->kind == Obj::TORCH) ? o->payload : o;
    if (!t || t->kind != Obj::TABLE) return nullptr;
    for (auto &kv : t->items)
        if (kv.first && kv.first->kind == Obj::STR && kv.first->str == name) return kv.second;
    return nullptr;
}

std::vector<Ref> array_of(const Ref &t)   // Lua array part 1..n in order
{
    std::vector<Ref> out;
    if (!t || t->kind != Obj::TABLE) return out;
    std::map<long long, Ref> byidx;
    for (auto &kv : t->items)
        if (kv.first && kv.first->kind == Obj::NUM) byidx[(long long)kv.first->num] = kv.second;
    for (long long i = 1;; ++i) {
        auto it = byidx.find(i);
        if (it == byidx.end()) break;
        out.push_back(it->second);
    }
    return out;
}

bool class_is(const Ref &o, const char *suffix)
{
    if (!o || (o->kind != Obj::TORCH)) return false;
    const size_t n = strlen(suffix);
    return o->str.size() >= n && o->str.compare(o->str.size() - n, n, suffix) == 0;
}

bool tensor_to_vector(const Ref &t, std::vector<float> &out)
{
    out.clear();
    if (!t || t->kind != Obj::TENSOR || !t->storage || t->storage->kind != Obj::STORAGE) return false;
    long long n = 1;
    for (long long s : t->size) n *= s;
    if (t->size.empty()) n = 0;
    out.resize((size_t)n);
    const std::vector<float> &st = t->storage->data;
    std::vector<long long> idx(t->size.size(), 0);
    for (long long i = 0; i < n; ++i) {
        long long off = t->offset;
        for (size_t d = 0; d < idx.size(); ++d) off += idx[d] * t->stride[d];
        if (off < 0 || off >= (long long)st.size()) return false;
        out[(size_t)i] = st[(size_t)off];
        for (int d = (int)idx.size() - 1; d >= 0; --d) {
            if (++idx[d] < t->size[d]) break;
            idx[d] = 0;
        }
    }
    return true;
}

Ref find_gmodule(const Ref &o, std::set<const Obj *> &visited, int depth = 0)
{
    if (!o || depth > 64 || visited.count(o.get())) return nullptr;
    visited.insert(o.get());
    if (class_is(o, "nn.gModule")) return o;
    const Ref &t = (o->kind == Obj::TORCH) ? o->payload : o;
    if (!t || t->kind != Obj::TABLE) return nullptr;
    // DataParallelTable keeps its replicas in `modules` (util.lua:50-58); search the fields
    for (auto &kv : t->items) {
        if (!kv.second) continue;
        if (kv.second->kind == Obj::TORCH || kv.second->kind == Obj::TABLE) {
            Ref r = find_gmodule(kv.second, visited, depth + 1);
            if (r) return r;
        }
    }
    return nullptr;
}

struct ConvW {
    int ci = 0, co = 0, stride = 1;
    std::vector<float> w, b;
};

bool read_conv(const Ref &m, ConvW &c, std::string &err)
{
    Ref ni = field(m, "nInputPlane"), no = field(m, "nOutputPlane"), dw = field(m, "dW");
    if (!ni || !no) { err = "SpatialConvolution without nInputPlane/nOutputPlane"; return false; }
    c.ci = (int)ni->num; c.co = (int)no->num; c.stride = dw ? (int)dw->num : 1;
    if (!tensor_to_vector(field(m, "weight"), c.w) || !tensor_to_vector(field(m, "bias"), c.b)) {
        err = "SpatialConvolution without readable weight/bias"; return false;
    }
    if ((long long)c.w.size() != (long long)c.co * c.ci * 9 || (int)c.b.size() != c.co) {
        err = "SpatialConvolution weight is not Co x Ci x 3 x 3"; return false;
    }
    return true;
}

// sign summary of the nn.MulConstant nodes reachable from `node` through up-sampling / identity
// nodes (depth-limited): bit0 = a positive constant seen, bit1 = a negative one.
int mulconstant_signs(const Ref &node, int depth)
{
    if (!node || depth > 6) return 0;
    int s = 0;
    for (const Ref &ch : array_of(field(node, "children"))) {
        Ref data = field(ch, "data");
        Ref mod = data ? field(data, "module") : nullptr;
        if (!mod) continue;
        if (class_is(mod, "MulConstant")) {
            Ref k = field(mod, "constant_scalar");
            if (k && k->num > 0) s |= 1;
            if (k && k->num < 0) s |= 2;
        } else if (class_is(mod, "SpatialUpSamplingBilinear") || class_is(mod, "Identity")) {
            s |= mulconstant_signs(ch, depth + 1);
        }
    }
    return s;
}

}  // namespace

// Reads a model file into the canonical flat order of graph `g`.  infer = true: the graph shape is taken from the file -- win
// from the nn.CostVolMulti nodes' `win` field (CostVolMulti.lua:26-37), levels from the convUnits present, skip (pwc_skip) from
// the coarsest .. finest decoder levels found (pwc.lua:136,237), the remaining createModelMulti options at their defaults; infer =
// false: the file must BE graph `g` (same win, same units; the decoders' input widths are matched against g's, so two_frame /
// pwc_sum_cvs / occ_input files are told from the default wiring).  g.past_flow is set from the file either way.
bool load_t7_ex(const std::string &path, GraphOpts &g, bool infer, std::vector<float> &flat, std::string &err)
{
    Reader R;
    R.f = fopen(path.c_str(), "rb");
    if (!R.f) { err = "cannot open '" + path + "'"; return false; }
    Ref root = R.object();
    fclose(R.f);
    if (!R.ok) { err = "'" + path + "' is not a readable binary .t7 file: " + R.err; return false; }
    std::set<const Obj *> visited;
    Ref gm = find_gmodule(root, visited);   // unwraps nn.DataParallelTable, back2future.lua:114-116
    if (!gm) { err = "no nn.gModule found in '" + path + "'"; return false; }
    std::vector<Ref> nodes = array_of(field(gm, "forwardnodes"));
    if (nodes.empty()) { err = "gModule has no forwardnodes"; return false; }

    static const int kLevelOfFeatIn[8] = {0, 0, 3, 16, 32, 64, 96, 128};   // nInputPlane of convUnit l (pwc_skip >= 1)
    struct Dec {
        int n, kind;
        int pos = 0;                                                      // index of the node in forwardnodes
        std::vector<ConvW> convs;
    };
    std::map<int, std::vector<ConvW>> feat;                               // level -> 2 convs
    std::vector<Dec> decs;
    std::set<const Obj *> seen_seq;
    int file_win = 0;
    int node_pos = -1;
    for (const Ref &node : nodes) {
        ++node_pos;
        Ref data = field(node, "data");
        Ref mod = data ? field(data, "module") : nullptr;
        if (mod && class_is(mod, "nn.CostVolMulti")) {
            Ref wn = field(mod, "win");
            if (wn && wn->num > 0) {
                // untrusted input: range-check the double before it becomes an int (GraphOpts::valid(): odd, 1..15)
                const double wv = wn->num;
                if (!(wv >= 1.0 && wv <= 15.0) || wv != (double)(int)wv || !((int)wv & 1)) { err = "CostVolMulti window outside 1..15 / not an odd integer"; return false; }
                if (file_win && file_win != (int)wv) { err = "CostVolMulti nodes with different windows"; return false; }
                file_win = (int)wv;
            }
        }
        if (!mod || !class_is(mod, "nn.Sequential") || seen_seq.count(mod.get())) continue;
        seen_seq.insert(mod.get());
        std::vector<ConvW> convs;
        for (const Ref &sub : array_of(field(mod, "modules"))) {
            if (!class_is(sub, "SpatialConvolution")) continue;
            ConvW c;
            if (!read_conv(sub, c, err)) return false;
            convs.push_back(std::move(c));
        }
        if (convs.size() == 2) {
            int level = 0;
            for (int l = 2; l <= 7; ++l)
                if (convs[0].ci == kLevelOfFeatIn[l] && convs[0].co == kFeatH[l] && convs[0].stride == 2) level = l;
            // pwc_skip = 0 (pwc.lua:120-122,171-173): convUnit(3, 16, 1) on level 1, convUnit(16, 16, 2) on level 2
            if (convs[0].ci == 3 && convs[0].co == kFeatH[2] && convs[0].stride == 1) level = 1;
            if (convs[0].ci == kFeatH[2] && convs[0].co == kFeatH[2] && convs[0].stride == 2) level = 2;
            if (!level) { err = "unexpected convUnit shape"; return false; }
            if (!feat.count(level)) feat[level] = std::move(convs);   // the three siamese clones share weights
        } else if (convs.size() == 6) {
            Dec d;
            d.n = convs[0].ci;
            bool to_softmax = false;
            for (const Ref &ch : array_of(field(node, "children"))) {
                Ref d2 = field(ch, "data");
                Ref m2 = d2 ? field(d2, "module") : nullptr;
                if (m2 && class_is(m2, "SpatialSoftMax")) to_softmax = true;
            }
            if (to_softmax) d.kind = KIND_OCC;
            else d.kind = (mulconstant_signs(node, 0) & 1) ? KIND_FLOW : KIND_PAST;
            d.convs = std::move(convs);
            d.pos = node_pos;
            decs.push_back(std::move(d));
        }
    }
    if (feat.empty() && !decs.empty()) { err = "no convUnit in the graph: a pwc_siamese = 0 model's decoders cannot be told apart by their shapes -- not supported from .t7 (pass the flat weights to b2f_init_ex instead)"; return false; }
    if (feat.empty() || decs.empty()) { err = "no convUnit / decoder found in the graph"; return false; }
    if (!infer && !g.siamese) { err = "pwc_siamese = 0 graphs are not supported from .t7"; return false; }
    if (!infer && (feat.count(1) != 0) != (g.skip == 0)) { err = std::string("the file ") + (feat.count(1) ? "has" : "has no") + " level-1 convUnit, the graph options say pwc_skip = " + std::to_string(g.skip); return false; }
    if (infer) {
        g = GraphOpts();
        if (file_win) g.win = file_win;
        if (feat.count(1)) g.skip = 0;   // featMaps[1] = 16 enters occ_in() / flow_in() of level 1 below
        g.levels = feat.rbegin()->first;
        // win / levels in range before they enter occ_in() / flow_in() (win * win) below; skip is inferred later
        if (!(g.win >= 1 && (g.win & 1) && g.win <= 15 && g.levels >= 2 && g.levels <= 7)) { err = "the file's graph shape is outside what this library runs: " + graph_opts_string(g); return false; }
    } else if (file_win && file_win != g.win) {
        err = "the file's cost volumes use a " + std::to_string(file_win) + "-wide window, the graph options say " + std::to_string(g.win);
        return false;
    }
    if (feat.rbegin()->first != g.levels) { err = "the file has " + std::to_string(feat.rbegin()->first) + " pyramid levels, the graph options say " + std::to_string(g.levels); return false; }
    // decoder level = the level whose first-layer width for that role is the decoder's (widths differ per level: pwc.lua:288-337)
    // With pwc_skip = 0 levels 1 and 2 both carry 16 maps (pwc.lua:120-122): their decoders have the same width and are told
    // apart by the order of the nodes -- forwardnodes is a topological order (level 2 feeds level 1) or, in files written
    // back to front, its reverse, which the position of the coarsest level's decoder (width = the cost volume alone) shows.
    std::map<std::pair<int, int>, std::vector<ConvW>> dec;                // (level, kind) -> 6 convs
    int lmin = 8, pos_coarsest = -1, pos_other = -1;
    for (const Dec &d : decs) {
        if (d.kind == KIND_OCC) continue;
        if (d.n == g.flow_in(g.levels)) pos_coarsest = d.pos;
        else pos_other = d.pos;                                            // any finer level's flow decoder
    }
    const bool back_to_front = pos_coarsest >= 0 && pos_other >= 0 && pos_coarsest > pos_other;
    for (Dec &d : decs) {
        int level = 0;
        for (int l = g.feat_first(); l <= g.levels; ++l)
            if (d.n == (d.kind == KIND_OCC ? g.occ_in(l) : g.flow_in(l))) {
                if (level == 1 && l == 2 && g.skip == 0) {                 // fits both 16-map levels: decide by the order
                    int partner = -1;
                    for (const Dec &e : decs)
                        if (&e != &d && e.kind == d.kind && e.n == d.n) partner = e.pos;
                    if (partner < 0) { err = "one decoder for the two 16-map levels"; return false; }
                    const bool earlier = back_to_front ? d.pos > partner : d.pos < partner;
                    level = earlier ? 2 : 1;
                    continue;
                }
                if (level) { err = "decoder input width " + std::to_string(d.n) + " fits two levels"; return false; }
                level = l;
            }
        if (!level) { err = "unexpected decoder input width " + std::to_string(d.n) + " (window / two_frame / pwc_sum_cvs / occ_input of the file differ from the graph options?)"; return false; }
        if (dec.count({level, d.kind})) { err = "two decoders with the same role at level " + std::to_string(level); return false; }
        dec[{level, d.kind}] = std::move(d.convs);
        lmin = std::min(lmin, level);
    }
    if (infer && g.skip == 0 && lmin != 1) { err = "level-1 convUnit but no level-1 decoder"; return false; }
    if (infer) g.skip = lmin - 1;
    else if (lmin != g.l_st()) { err = "the file's finest decoder level is " + std::to_string(lmin) + ", the graph options say " + std::to_string(g.l_st()); return false; }
    bool past_flow = false;
    for (auto &kv : dec)
        if (kv.first.second == KIND_PAST) past_flow = true;
    Ref pf = field(gm, "past_flow");   // model.past_flow, pwc.lua:494
    if (pf && pf->kind == Obj::BOOL && pf->b != past_flow) { err = "past_flow field disagrees with the graph"; return false; }
    g.past_flow = past_flow;
    if (!g.valid()) { err = "the file's graph shape is outside what this library runs: " + graph_opts_string(g); return false; }

    long long total = 0;
    const std::vector<ConvDesc> lay = weight_layout(g, &total);
    flat.assign((size_t)total, 0.f);
    for (const ConvDesc &d : lay) {
        const ConvW *src = nullptr;
        if (d.kind == KIND_FEAT) {
            auto it = feat.find(d.level);
            if (it != feat.end()) src = &it->second[(size_t)d.idx - 1];
        } else {
            auto it = dec.find({d.level, d.kind});
            if (it != dec.end()) src = &it->second[(size_t)d.idx - 1];
        }
        if (!src) { err = "model is missing a convolution (level " + std::to_string(d.level) + ")"; return false; }
        if (src->ci != d.ci || src->co != d.co) { err = "convolution shape mismatch at level " + std::to_string(d.level); return false; }
        memcpy(flat.data() + d.w_off, src->w.data(), src->w.size() * sizeof(float));
        memcpy(flat.data() + d.b_off, src->b.data(), src->b.size() * sizeof(float));
    }
    return true;
}

// the shipped graph shape (win 9, levels 7, skip 2), as back2future.lua:97-113 loads it
bool load_t7(const std::string &path, std::vector<float> &flat, bool &past_flow, std::string &err)
{
    GraphOpts g;
    if (!load_t7_ex(path, g, false, flat, err)) return false;
    past_flow = g.past_flow;
    return true;
}

}  // namespace b2f
