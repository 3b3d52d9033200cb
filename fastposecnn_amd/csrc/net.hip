// net.hip — the backbone engine behind fpc_net_* (include/fpc.h): a native plan of
// PoseRegressor.pure_model_forward + Model.class_compression
// (F/lib/pose_regressor.py:709-743, 445-457) for inference.
//
// The graph is the one segmentation_models_pytorch builds for FastPoseCNN
// (encoder = ResNet BasicBlock x {2,2,2,2} or {3,4,6,3}; four FPN decoders, merge "add";
// four 1x1 heads + x4 bilinear), see fastposecnn_amd/lib/backbone.py.  The plan owns no device
// memory: packed weights, activations and split-K scratch live in one caller-provided workspace.
// Launch order per frame (R18): stem conv, max-pool, 16 encoder convs (+3 downsample 1x1) with
// BatchNorm / residual / ReLU in their epilogues, then the FOUR decoders as grouped launches:
// 4 lateral 1x1 convs (FPN top-down add in the epilogue), 7 3x3 convs (GroupNorm partial sums in
// the epilogue), 7 GroupNorm finalisations, 3 GN+ReLU+x2-upsample passes, 1 merge+head kernel,
// 1 x4-upsample + class-compression kernel.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include "net_kernels.hpp"

namespace fpc {

struct PackedConv {
    int Cin = 0, Cinp = 0, Cout = 0, Kh = 1, Kw = 1, stride = 1, pad = 0;   // Cinp: channels of the input LAYOUT
    int Kwp = 1;             // taps per kernel row in the packed layout (= Kw; 8 for the stem's row-per-K-step form)
    int K = 0, Kpad = 0, Npad = 0;
    int p_w = -1;            // parameter index of the OIHW weight
    int p_bn = -1;           // first of (weight, bias, running_mean, running_var) or -1
    int p_bias = -1;         // conv bias or -1
    size_t w_off = 0, scale_off = 0, shift_off = 0;   // float offsets into the workspace
    size_t wino_off = 0;     // Winograd-packed copy (eligible convs only)
    bool wino_ok = false;    // 3x3, stride 1, pad 1, Cin % 8 == 0, Cout % 64 == 0
};

struct Act { size_t off = 0; int H = 0, W = 0, C = 0; };

struct ConvPlan {
    int bm = 64, bn = 64, nsplit = 1, mtiles = 1, ntiles = 1, wino = 0, bf3 = 0, fused = 0;
    int lat = 0;             // > 0: k_lateral1x1 with this many workgroups per 128-pixel tile (lateral.hip) instead of k_conv_igemm
    int stem = 0;            // > 0: k_stem7x7 (stem.hip), a persistent grid of this many workgroups (one per CU: 256)
};

// the 7x7 / stride-2 / pad-3 stem in its row-per-K-step layout (NHWC4 image, 8 taps x 4 channels per kernel row, K = 224) with a
// BatchNorm / bias-only epilogue and an output row that divides into 64-pixel segments
static bool stem_ok(const ConvArgs& a) {
    return a.lanepx == 1 && a.generic == 0 && a.Kh == 7 && a.stride == 2 && a.pad == 3 && a.Cout == 64 && a.Kpad == 224 &&
           a.Wo % 64 == 0 && !a.p[0].res && !a.p[0].up && !a.p[0].gn_part && a.in_sc == 1 && a.in_sw == 4 &&
           a.in_sh == (long long)4 * a.Wi && a.in_sb == (long long)4 * a.Hi * a.Wi;
}

// a grouped 1x1 / stride-1 site with K = 64 or 128, bias-only epilogue and one input shared by the groups: the FPN laterals
// of the stride-4 and stride-8 maps (and fpc_conv2d's test hook)
static bool lateral_ok(const ConvArgs& a, int groups) {
    if (a.Kh != 1 || a.Kw != 1 || a.stride != 1 || a.pad != 0 || a.generic != 0 || (a.Cin != 64 && a.Cin != 128) || a.Kpad != a.Cin ||
        a.Cout % 32 != 0 || a.in_sc != 1 || a.in_sw != a.Cin || a.in_sh != (long long)a.Wi * a.Cin ||
        a.in_sb != (long long)a.Hi * a.Wi * a.Cin || a.lanepx)
        return false;
    for (int g = 0; g < groups; ++g)
        if (a.p[g].in != a.p[0].in || a.p[g].scale || a.p[g].res || a.p[g].gn_part || ((a.p[g].up != nullptr) != (a.p[0].up != nullptr)))
            return false;
    return !(a.p[0].up && ((a.Ho | a.Wo) & 1));
}

// fused split-K (ConvArgs::fused) needs one arrival counter per output tile
static bool can_fuse(const ConvPlan& p, int groups, int B) {
    return p.nsplit > 1 && (long long)groups * B * p.mtiles * p.ntiles <= kConvTickets;
}

static ConvPlan plan_conv(int HoWo, int B, int Cout, int ksteps, int groups, int force_bm = 0, int force_bn = 0,
                          int force_split = 0) {
    ConvPlan best;
    double best_t = 1e300;
    const int bms[2] = {64, 128}, bns[2] = {64, 128};
    for (int bi = 0; bi < 2; ++bi)
        for (int bj = 0; bj < 2; ++bj) {
            int bm = bms[bi], bn = bns[bj];
            if (force_bm && bm != force_bm) continue;
            if (force_bn && bn != force_bn) continue;
            if (!force_bn && bn == 128 && Cout <= 64) continue;
            int mt = cdiv(HoWo, bm), nt = cdiv(Cout, bn);
            for (int ns = 1; ns <= 32; ++ns) {
                if (force_split && ns != force_split) continue;
                int per = cdiv(ksteps, ns);
                if ((ns - 1) * per >= ksteps) continue;
                if (ns > 1 && (Cout % 4 != 0)) continue;
                double nblk = (double)groups * B * mt * nt * ns;
                double work = (double)(bm / 64) * (bn / 64) * per * 16.0 * 64.0 + 4000.0;
                double t = ceil(nblk / 256.0) * work + (ns > 1 ? 10000.0 + 200.0 * ns : 0.0);
                if (t < best_t) { best_t = t; best = ConvPlan{bm, bn, ns, mt, nt, 0}; }
            }
        }
    best.fused = can_fuse(best, groups, B) ? 1 : 0;
    return best;
}

// Tilings the autotuner may try for one convolution site (the heuristic plan is always among them).
static std::vector<ConvPlan> conv_candidates(int HoWo, int B, int Cout, int ksteps, int groups) {
    std::vector<ConvPlan> out;
    const int splits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    for (int bm = 64; bm <= 128; bm += 64)
        for (int bn = 64; bn <= 128; bn += 64) {
            if (bn == 128 && Cout <= 64) continue;
            int mt = cdiv(HoWo, bm), nt = cdiv(Cout, bn);
            long long base = (long long)groups * B * mt * nt;
            for (int ns : splits) {
                int per = cdiv(ksteps, ns);
                if (ns > 1 && (per < 2 || (ns - 1) * per >= ksteps || Cout % 4 != 0)) continue;
                if (ns > 1 && base * ns > 4096) continue;          // already plenty of workgroups
                out.push_back(ConvPlan{bm, bn, ns, mt, nt, 0});
                if (can_fuse(out.back(), groups, B)) { ConvPlan f = out.back(); f.fused = 1; out.push_back(f); }
            }
        }
    return out;
}

static size_t splitk_floats_for(const ConvPlan& p, int groups, int B, int Npad) {
    return p.nsplit > 1 ? (size_t)groups * p.nsplit * B * p.mtiles * p.bm * Npad : 0;
}

}  // namespace fpc

using namespace fpc;

struct fpc_net {
    int layers[4];
    int classes, B, H, W;
    bool r34 = false;
    std::vector<std::string> pnames;
    std::vector<int64_t> pnumel;
    std::vector<const float*> pptr;
    std::vector<PackedConv> convs;
    size_t packed_floats = 0;     // packed weights + folded BN region (persistent across forwards)
    size_t total_floats = 0;      // + activations and scratch
    float* ws = nullptr;
    bool loaded = false;
    bool tuning = false;          // next forward times every candidate tiling per conv site and keeps the best
    bool tuned = false;
    int tune_mode = 0;            // 0: minimise latency, 1: latency x sqrt(share of the chip occupied)

    // conv indices
    int c_stem = -1;
    struct Block { int conv1, conv2, ds; };
    std::vector<Block> blocks[4];
    struct Dec {
        int lat[4];               // p5, p4, p3, p2 (1x1, bias)
        int seg[7];               // s5.0 s5.1 s5.2 s4.0 s4.1 s3.0 s2.0
        int p_gn[7];              // param index of GN weight (bias = +1)
        int p_head_w, p_head_b;
        int head_ch, head_chp;
    } dec[4];

    // activations (float offsets)
    Act a_img4, a_stem, a_pool;
    std::vector<Act> a_blk_t[4], a_blk_y[4], a_blk_d[4];
    Act a_p[4][4];                // [decoder][p5,p4,p3,p2]
    Act a_seg[4][7];              // pre-GroupNorm conv outputs
    Act a_up[4][3];               // s5.0 -> up, s5.1 -> up, s4.0 -> up
    size_t gn_part_off[4][7], gn_aff_off[4][7];
    int gn_P[7];
    Act a_low[4];                 // low-res logits
    Act a_lsum[4];                // two-pass merge + head: the head of the three upsampled branches' sum at their own resolution
    int merge_split = -1;         // -1: two passes unless FPC_MERGE_SPLIT=0, 0: k_merge_head (one pass), 1: two passes
    size_t splitk_off = 0, splitk_floats = 0;
    int use_graph = 0;            // replay the frame-invariant launches as a HIP graph (fpc_net_set_graph)
    int split_precision = 0;      // autotuning may pick the bf16 x 3 form of a direct convolution (fpc_net_set_split_precision)
    hipGraphExec_t graph_exec = nullptr;
    size_t zeros_off = 0;         // 64 zero floats (DMA source for out-of-image positions)
    size_t tickets_off = 0;       // kConvTickets zero ints: arrival counters of the fused split-K convolutions

    // per-conv launch plans (index = conv id of decoder 0 for grouped ones)
    std::vector<ConvPlan> cplan;
    std::vector<int> c_howo, c_groups;   // output pixels per image and launch multiplicity of every planned conv site (0: not a site)

    size_t bump = 0;
    size_t alloc(size_t n) { size_t o = bump; bump += (n + 63) / 64 * 64; return o; }
    Act alloc_act(int h, int w, int c) { Act a; a.H = h; a.W = w; a.C = c; a.off = alloc((size_t)B * h * w * c); return a; }
    int add_param(const std::string& n, int64_t numel) { pnames.push_back(n); pnumel.push_back(numel); return (int)pnames.size() - 1; }

    int add_conv(const std::string& wname, int Cin, int Cout, int k, int stride, int pad, const char* bn_prefix,
                 const char* bias_name, int Cinp = 0, int Kwp = 0) {
        PackedConv c;
        c.Cin = Cin; c.Cinp = Cinp ? Cinp : Cin; c.Cout = Cout; c.Kh = c.Kw = k; c.stride = stride; c.pad = pad;
        c.Kwp = Kwp ? Kwp : k;
        c.K = c.Cinp * k * c.Kwp; c.Kpad = cdiv(c.K, kConvBK) * kConvBK; c.Npad = cdiv(Cout, kConvNAlign) * kConvNAlign;
        c.p_w = add_param(wname, (int64_t)Cout * Cin * k * k);
        if (bn_prefix) {
            std::string p(bn_prefix);
            c.p_bn = add_param(p + ".weight", Cout);
            add_param(p + ".bias", Cout);
            add_param(p + ".running_mean", Cout);
            add_param(p + ".running_var", Cout);
        }
        if (bias_name) c.p_bias = add_param(bias_name, Cout);
        c.w_off = alloc(conv_packed_floats(c.Npad, c.Kpad));      // f32 image + its three bf16 planes
        if (bn_prefix) { c.scale_off = alloc(Cout); c.shift_off = alloc(Cout); }
        c.wino_ok = (k == 3 && stride == 1 && pad == 1 && c.Cinp == Cin && Cin % 8 == 0 && Cout % 64 == 0);
        // f32 image (16 x) + split-precision image (24 x) + the 128-channel form's fragment-order image (24 x, wino128.hip)
        // ... + the fp16 x 2 form's fragment-order image (16 x + its two-float tail, wino_h2.hip)
        if (c.wino_ok) c.wino_off = alloc((size_t)(Cout % 128 == 0 ? 96 : 72) * Cout * Cin + 80);
        convs.push_back(c);
        return (int)convs.size() - 1;
    }
};

static const char* kDecNames[4] = {"mask_decoder", "rotation_decoder", "translation_decoder", "scales_decoder"};
static const char* kHeadNames[4] = {"segmentation_head", "rotation_head", "translation_head", "scales_head"};

static int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

extern "C" int fpc_net_create(const char* encoder, int classes, int B, int H, int W, fpc_net_t** out) {
    if (!encoder || !out || classes < 2 || classes > 8 || B < 1 || H < 32 || W < 32 || H % 32 || W % 32) return FPC_EINVAL;
    fpc_net* n = new fpc_net();
    if (!strcmp(encoder, "resnet18")) { int l[4] = {2, 2, 2, 2}; memcpy(n->layers, l, sizeof(l)); }
    else if (!strcmp(encoder, "resnet34")) { int l[4] = {3, 4, 6, 3}; memcpy(n->layers, l, sizeof(l)); n->r34 = true; }
    else { delete n; return FPC_EINVAL; }
    n->classes = classes; n->B = B; n->H = H; n->W = W;
    {   // merge + head in two passes (merge_split.hip) unless FPC_MERGE_SPLIT=0; read once, here
        const char* e = getenv("FPC_MERGE_SPLIT");
        n->merge_split = e ? (atoi(e) != 0) : 1;      // (one frame: 27 us in two passes against 32.5 us in one)
    }
    char buf[256];

    // ---- parameters + packed storage (persistent region first)
    n->c_stem = n->add_conv("encoder.conv1.weight", 3, 64, 7, 2, 3, "encoder.bn1", nullptr, 4, 8);     // NHWC4 pixels, 8 taps per row
    const int planes[4] = {64, 128, 256, 512};
    int inpl = 64;
    for (int L = 0; L < 4; ++L)
        for (int bi = 0; bi < n->layers[L]; ++bi) {
            int stride = (bi == 0 && L > 0) ? 2 : 1;
            fpc_net::Block blk;
            snprintf(buf, sizeof(buf), "encoder.layer%d.%d", L + 1, bi);
            std::string p(buf);
            blk.conv1 = n->add_conv(p + ".conv1.weight", inpl, planes[L], 3, stride, 1, (p + ".bn1").c_str(), nullptr);
            blk.conv2 = n->add_conv(p + ".conv2.weight", planes[L], planes[L], 3, 1, 1, (p + ".bn2").c_str(), nullptr);
            blk.ds = -1;
            if (stride != 1 || inpl != planes[L])
                blk.ds = n->add_conv(p + ".downsample.0.weight", inpl, planes[L], 1, stride, 0, (p + ".downsample.1").c_str(), nullptr);
            inpl = planes[L];
            n->blocks[L].push_back(blk);
        }
    const int G = classes - 1;
    const int head_ch[4] = {classes, 4 * G, 3 * G, 3 * G};
    for (int d = 0; d < 4; ++d) {
        std::string D(kDecNames[d]);
        fpc_net::Dec& dc = n->dec[d];
        dc.lat[0] = n->add_conv(D + ".p5.weight", 512, 256, 1, 1, 0, nullptr, (D + ".p5.bias").c_str());
        const int skipc[3] = {256, 128, 64};
        for (int i = 0; i < 3; ++i) {
            snprintf(buf, sizeof(buf), "%s.p%d.skip_conv", kDecNames[d], 4 - i);
            std::string p(buf);
            dc.lat[1 + i] = n->add_conv(p + ".weight", skipc[i], 256, 1, 1, 0, nullptr, (p + ".bias").c_str());
        }
        const int nconv[4] = {3, 2, 1, 1};
        int si = 0;
        for (int sb = 0; sb < 4; ++sb)
            for (int j = 0; j < nconv[sb]; ++j) {
                snprintf(buf, sizeof(buf), "%s.seg_blocks.%d.block.%d.block", kDecNames[d], sb, j);
                std::string p(buf);
                dc.seg[si] = n->add_conv(p + ".0.weight", j == 0 ? 256 : 128, 128, 3, 1, 1, nullptr, nullptr);
                dc.p_gn[si] = n->add_param(p + ".1.weight", 128);
                n->add_param(p + ".1.bias", 128);
                ++si;
            }
    }
    for (int d = 0; d < 4; ++d) {
        fpc_net::Dec& dc = n->dec[d];
        dc.head_ch = head_ch[d];
        dc.head_chp = (head_ch[d] + 3) / 4 * 4;
        dc.p_head_w = n->add_param(std::string(kHeadNames[d]) + ".0.weight", (int64_t)head_ch[d] * 128);
        dc.p_head_b = n->add_param(std::string(kHeadNames[d]) + ".0.bias", head_ch[d]);
    }
    n->zeros_off = n->alloc(64);
    n->tickets_off = n->alloc(kConvTickets);
    n->packed_floats = n->bump;

    // ---- activations
    int h1 = conv_out(H, 7, 2, 3), w1 = conv_out(W, 7, 2, 3);
    n->a_img4 = n->alloc_act(H, W, 4);
    n->a_stem = n->alloc_act(h1, w1, 64);
    int hp = conv_out(h1, 3, 2, 1), wp = conv_out(w1, 3, 2, 1);
    n->a_pool = n->alloc_act(hp, wp, 64);
    int fh[4], fw[4];
    {
        int h = hp, w = wp;
        for (int L = 0; L < 4; ++L) {
            if (L > 0) { h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1); }
            fh[L] = h; fw[L] = w;
            for (int bi = 0; bi < n->layers[L]; ++bi) {
                n->a_blk_t[L].push_back(n->alloc_act(h, w, planes[L]));
                n->a_blk_y[L].push_back(n->alloc_act(h, w, planes[L]));
                n->a_blk_d[L].push_back(n->blocks[L][bi].ds >= 0 ? n->alloc_act(h, w, planes[L]) : Act());
            }
        }
    }
    // FPN needs exact x2 relations between levels
    for (int L = 1; L < 4; ++L)
        if (fh[L - 1] != 2 * fh[L] || fw[L - 1] != 2 * fw[L]) { delete n; return FPC_EINVAL; }
    // seg conv geometry: index -> (resolution level): s5.0@L3, s5.1@L2, s5.2@L1, s4.0@L2, s4.1@L1, s3.0@L1, s2.0@L0
    const int seg_level[7] = {3, 2, 1, 2, 1, 1, 0};
    for (int d = 0; d < 4; ++d) {
        for (int i = 0; i < 4; ++i) n->a_p[d][i] = n->alloc_act(fh[3 - i], fw[3 - i], 256);
        for (int i = 0; i < 7; ++i) n->a_seg[d][i] = n->alloc_act(fh[seg_level[i]], fw[seg_level[i]], 128);
        n->a_up[d][0] = n->alloc_act(fh[2], fw[2], 128);   // up2(s5.0)
        n->a_up[d][1] = n->alloc_act(fh[1], fw[1], 128);   // up2(s5.1)
        n->a_up[d][2] = n->alloc_act(fh[1], fw[1], 128);   // up2(s4.0)
        n->a_low[d] = n->alloc_act(fh[0], fw[0], n->dec[d].head_chp);
        n->a_lsum[d] = n->alloc_act(fh[1], fw[1], n->dec[d].head_chp);
    }

    // ---- conv plans (+ split-K scratch for the worst candidate, GroupNorm partials for the largest P32)
    n->cplan.resize(n->convs.size());
    n->c_howo.assign(n->convs.size(), 0);
    n->c_groups.assign(n->convs.size(), 0);
    auto plan = [&](int ci, int HoWo, int groups) {
        const PackedConv& c = n->convs[ci];
        n->c_howo[ci] = HoWo; n->c_groups[ci] = groups;
        n->cplan[ci] = plan_conv(HoWo, B, c.Cout, c.Kpad / kConvBK, groups);
        size_t need = splitk_floats_for(n->cplan[ci], groups, B, c.Npad);
        for (const ConvPlan& q : conv_candidates(HoWo, B, c.Cout, c.Kpad / kConvBK, groups)) {
            size_t f = splitk_floats_for(q, groups, B, c.Npad);
            if (f * sizeof(float) <= ((size_t)256 << 20) && f > need) need = f;
        }
        if (need > n->splitk_floats) n->splitk_floats = need;
    };
    plan(n->c_stem, h1 * w1, 1);
    for (int L = 0; L < 4; ++L)
        for (auto& blk : n->blocks[L]) {
            plan(blk.conv1, fh[L] * fw[L], 1);
            plan(blk.conv2, fh[L] * fw[L], 1);
            if (blk.ds >= 0) plan(blk.ds, fh[L] * fw[L], 1);
        }
    for (int i = 0; i < 4; ++i) plan(n->dec[0].lat[i], fh[3 - i] * fw[3 - i], 4);
    for (int i = 0; i < 7; ++i) {
        int HoWo = fh[seg_level[i]] * fw[seg_level[i]];
        plan(n->dec[0].seg[i], HoWo, 4);
        n->gn_P[i] = cdiv(HoWo, 128) * 4;      // upper bound of mtiles*bm/32 over the tilings
        for (int d = 0; d < 4; ++d) {
            n->gn_part_off[d][i] = n->alloc((size_t)B * n->gn_P[i] * 128 * 2);
            n->gn_aff_off[d][i] = n->alloc((size_t)B * 128 * 2);
        }
    }
    n->splitk_off = n->alloc(n->splitk_floats);
    n->total_floats = n->bump;
    n->pptr.assign(n->pnames.size(), nullptr);
    *out = n;
    return FPC_OK;
}

extern "C" void fpc_net_destroy(fpc_net_t* n) {
    if (!n) return;
    if (n->graph_exec) (void)hipGraphExecDestroy(n->graph_exec);
    delete n;
}

// 1: after autotuning, the ~57 launches of a frame that only touch the plan's workspace are captured once and
// replayed with one hipGraphLaunch per frame (the first and the last kernel take the caller's tensors and stay
// ordinary launches).  Changing the tilings (fpc_net_autotune_next) or the parameters drops the recorded graph.
// 1: the next autotuning pass also times the split-precision (bf16 x 3, f32 accumulation) form of every direct
// fast-path convolution and keeps it where it is faster.  Opt-in: the results then differ from the f32 product chain
// by rounding (about 2^-24 relative per product, like a different f32 summation order), not bit for bit.
extern "C" int fpc_net_set_split_precision(fpc_net_t* n, int on) {
    if (!n) return FPC_EINVAL;
    n->split_precision = on < 0 ? 0 : (on > 3 ? 3 : on);      // 0: f32 products only, 1: + bf16 x 3 forms, 2: + the fp16 x 2 Winograd form, 3: + its three-product form
    return FPC_OK;
}

extern "C" int fpc_net_set_graph(fpc_net_t* n, int on) {
    if (!n) return FPC_EINVAL;
    n->use_graph = on ? 1 : 0;
    if (!on && n->graph_exec) { (void)hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; }
    return FPC_OK;
}
extern "C" int fpc_net_param_count(const fpc_net_t* n) { return n ? (int)n->pnames.size() : 0; }
extern "C" const char* fpc_net_param_name(const fpc_net_t* n, int i) {
    return (n && i >= 0 && i < (int)n->pnames.size()) ? n->pnames[i].c_str() : nullptr;
}
extern "C" int64_t fpc_net_param_numel(const fpc_net_t* n, int i) {
    return (n && i >= 0 && i < (int)n->pnumel.size()) ? n->pnumel[i] : -1;
}
extern "C" size_t fpc_net_workspace_bytes(const fpc_net_t* n) { return n ? n->total_floats * sizeof(float) : 0; }

extern "C" int fpc_net_load_params(fpc_net_t* n, const float* const* params, int count, void* ws, size_t ws_bytes,
                                   fpc_stream_t stream) {
    if (!n || !params || count != (int)n->pnames.size() || !ws) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0 || ws_bytes < n->total_floats * sizeof(float)) return FPC_EWORKSPACE;
    if (n->graph_exec) { (void)hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; }       // pointers may change
    for (int i = 0; i < count; ++i)
        if (!params[i] || ((uintptr_t)params[i] & 15)) return FPC_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    n->ws = (float*)ws;
    n->pptr.assign(params, params + count);
    if (hipMemsetAsync(n->ws + n->zeros_off, 0, 64 * sizeof(float), s) != hipSuccess) return FPC_ELAUNCH;
    if (hipMemsetAsync(n->ws + n->tickets_off, 0, kConvTickets * sizeof(int), s) != hipSuccess) return FPC_ELAUNCH;
    for (const PackedConv& c : n->convs) {
        int rc = launch_pack_weight(n->pptr[c.p_w], n->ws + c.w_off, c.Cout, c.Cin, c.Cinp, c.Kh, c.Kw, c.Kwp, c.Npad, c.Kpad, s);
        if (rc) return rc;
        rc = launch_pack_weight_bf3(n->pptr[c.p_w], n->ws + c.w_off, c.Cout, c.Cin, c.Cinp, c.Kh, c.Kw, c.Kwp, c.Npad, c.Kpad, s);
        if (rc) return rc;
        if (c.wino_ok) {
            rc = launch_wino_pack(n->pptr[c.p_w], n->ws + c.wino_off, c.Cout, c.Cin, s);
            if (rc) return rc;
            rc = launch_wino_pack_bf3(n->pptr[c.p_w], n->ws + c.wino_off + (size_t)16 * c.Cout * c.Cin, c.Cout, c.Cin, s);
            if (rc) return rc;
            if (c.Cout % 128 == 0) {
                rc = launch_wino_pack_c128(n->pptr[c.p_w], n->ws + c.wino_off + (size_t)40 * c.Cout * c.Cin, c.Cout, c.Cin, s);
                if (rc) return rc;
            }
            rc = launch_wino_pack_h2(n->pptr[c.p_w], n->ws + c.wino_off + (size_t)(c.Cout % 128 == 0 ? 64 : 40) * c.Cout * c.Cin, c.Cout, c.Cin, s);
            if (rc) return rc;
            if (c.Cin % 16 == 0) {      // the pair-order image of the three-product fp16 form, behind the fp16 x 2 image and its tail
                rc = launch_wino_pack_h3(n->pptr[c.p_w], n->ws + c.wino_off + (size_t)(c.Cout % 128 == 0 ? 80 : 56) * c.Cout * c.Cin + 8, c.Cout, c.Cin, s);
                if (rc) return rc;
            }
        }
        if (c.p_bn >= 0) {
            rc = launch_fold_bn(n->pptr[c.p_bn], n->pptr[c.p_bn + 1], n->pptr[c.p_bn + 2], n->pptr[c.p_bn + 3], 1e-5f,
                                c.Cout, n->ws + c.scale_off, n->ws + c.shift_off, s);
            if (rc) return rc;
        }
    }
    n->loaded = true;
    return FPC_OK;
}

namespace {

struct ConvIO {
    const float* in; long long sb, sh, sw, sc; int Hi, Wi;
    float* out; int Ho, Wo;
    const float* res; const float* up; float* gn_part;
};

// fills the shared part of ConvArgs from conv `c` + plan `p`
void fill_conv_args(const fpc_net* n, ConvArgs& a, const PackedConv& c, const ConvPlan& p, int Hi, int Wi, int Ho,
                    int Wo, long long sb, long long sh, long long sw, long long sc, bool relu, int mode) {
    memset(&a, 0, sizeof(a));
    a.B = n->B; a.Hi = Hi; a.Wi = Wi; a.Cin = c.Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = c.Cout; a.Npad = c.Npad;
    a.Kh = c.Kh; a.Kw = c.Kw; a.stride = c.stride; a.pad = c.pad; a.K = c.K; a.Kpad = c.Kpad;
    a.in_sb = sb; a.in_sh = sh; a.in_sw = sw; a.in_sc = sc;
    a.relu = relu ? 1 : 0; a.nsplit = p.nsplit; a.mtiles = p.mtiles; a.ntiles = p.ntiles; a.ksteps = c.Kpad / kConvBK;
    a.bm = p.bm; a.bn = p.bn; a.generic = mode;
    a.splitk_ws = n->ws + n->splitk_off;
    a.zeros = n->ws + n->zeros_off;
    a.tickets = (int*)(n->ws + n->tickets_off);
    a.fused = p.fused;
}

int launch_conv_plan(ConvArgs& a, const ConvPlan& p, int groups, hipStream_t s) {
    if (p.stem) {
        if (groups != 1 || !stem_ok(a)) return FPC_EINVAL;
        StemArgs t;
        memset(&t, 0, sizeof(t));
        t.in = a.p[0].in;
        t.wpl = reinterpret_cast<const unsigned short*>(a.p[0].w + (size_t)a.Npad * a.Kpad);      // the bf16 planes behind the f32 image
        t.out = a.p[0].out; t.scale = a.p[0].scale; t.shift = a.p[0].shift;
        t.B = a.B; t.Hi = a.Hi; t.Wi = a.Wi; t.Ho = a.Ho; t.Wo = a.Wo; t.Cout = a.Cout; t.Npad = a.Npad; t.Kpad = a.Kpad; t.relu = a.relu;
        t.grid = p.stem;
        return launch_stem7x7(t, s);
    }
    if (p.lat) {
        if (!lateral_ok(a, groups)) return FPC_EINVAL;
        LatArgs l;
        memset(&l, 0, sizeof(l));
        l.in = a.p[0].in;
        for (int g = 0; g < groups; ++g) {
            // the three bf16 planes sit behind the f32 image (k_pack_weight_bf3)
            l.wpl[g] = reinterpret_cast<const unsigned short*>(a.p[g].w + (size_t)a.Npad * a.Kpad);
            l.out[g] = a.p[g].out; l.shift[g] = a.p[g].shift; l.up[g] = a.p[g].up;
        }
        l.B = a.B; l.Ho = a.Ho; l.Wo = a.Wo; l.Cout = a.Cout; l.Npad = a.Npad; l.Kpad = a.Kpad; l.groups = groups; l.relu = a.relu;
        l.parts = p.lat;
        return launch_lateral1x1(l, s);
    }
    if (p.wino) {
        WinoArgs w;
        memset(&w, 0, sizeof(w));
        for (int g = 0; g < groups; ++g) {
            if (!a.wino_w[g] || a.p[g].up) return FPC_EINVAL;
            w.p[g] = a.p[g];
            // the split-precision image follows the f32 one, the 128-channel form's fragment-order image follows that
            w.p[g].w = a.wino_w[g] + ((p.wino == 5 || p.wino == 7) ? (size_t)16 * a.Cout * a.Cin : p.wino == 6 ? (size_t)40 * a.Cout * a.Cin
                                      : p.wino == 8 ? (size_t)(a.Cout % 128 == 0 ? 64 : 40) * a.Cout * a.Cin      // ... the fp16 x 2 image
                                      : p.wino == 9 ? (size_t)(a.Cout % 128 == 0 ? 80 : 56) * a.Cout * a.Cin + 8 : 0);      // ... its pair-order form last
        }
        w.variant = p.wino == 3 ? 1 : (p.wino == 4 ? 2 : (p.wino == 5 ? 3 : 0));
        w.zeros = a.zeros;
        w.dbg = (long long*)a.dbg;
        w.groups = groups;
        w.B = a.B; w.H = a.Ho; w.W = a.Wo; w.Cin = a.Cin; w.Cout = a.Cout; w.relu = a.relu;
        w.waves = (p.wino == 2 || p.wino == 4 || p.wino == 5 || p.wino == 7 || p.wino == 8 || p.wino == 9) ? 8 : 4;
        w.tbx = cdiv(cdiv(a.Wo, 2), 8); w.tby = cdiv(cdiv(a.Ho, 2), w.waves);
        if (p.wino == 6) return a.Cout % 128 == 0 ? launch_conv_wino_c128(w, groups, s) : FPC_EINVAL;      // 8 x 4 tiles x 128 channels (wino128.hip)
        if (p.wino == 7) return launch_conv_wino_w4(w, groups, s);
        if (p.wino == 9) return launch_conv_wino_h3(w, groups, s);      // three fp16 piece products over pairs of K-steps (wino_h3.hip)
        if (p.wino == 8) return launch_conv_wino_h2(w, groups, s);      // the same on two fp16 pieces per operand (wino_h2.hip)      // 8 x 8 tiles x 64 channels as four waves of 512 registers (wino_w4.hip)
        return launch_conv_wino(w, groups, s);
    }
    a.bm = p.bm; a.bn = p.bn; a.nsplit = p.nsplit; a.mtiles = p.mtiles; a.ntiles = p.ntiles; a.groups = groups;
    a.bf3 = (p.bf3 && a.generic == 0) ? 1 : 0;
    a.fused = (p.fused && p.nsplit > 1) ? 1 : 0;
    int rc = launch_conv(a, groups, s);
    if (rc) return rc;
    if (a.nsplit > 1 && !a.fused) rc = launch_conv_splitk_epilogue(a, groups, s);
    return rc;
}

// number of GroupNorm partial rows per image a plan writes
int plan_gn_rows(const ConvPlan& p, int Ho, int Wo) {
    return p.wino ? cdiv(cdiv(Wo, 2), 8) * cdiv(cdiv(Ho, 2), (p.wino == 2 || p.wino == 4 || p.wino == 5 || p.wino == 7 || p.wino == 8 || p.wino == 9) ? 8 : 4) : p.mtiles * p.bm / 32;
}

// Runs conv site `ci` with its current plan; in tuning mode first times every candidate tiling
// (HIP events on the stream, synchronising — only ever inside fpc_net_autotune) and keeps the fastest.
int run_conv(fpc_net* n, ConvArgs& a, int groups, int ci, hipStream_t s) {
    if (n && n->tuning) {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return FPC_ELAUNCH;
        float best_ms = 1e30f;
        ConvPlan best = n->cplan[ci];
        size_t cap = n->splitk_floats;
        std::vector<ConvPlan> cands = conv_candidates(a.Ho * a.Wo, a.B, a.Cout, a.ksteps, groups);
        if (n->split_precision && a.generic == 0) {          // the same tilings with split-precision matrix products
            size_t nc = cands.size();
            for (size_t i = 0; i < nc; ++i) { ConvPlan q = cands[i]; q.bf3 = 1; cands.push_back(q); }
        }
        if (n->split_precision && groups == 1 && stem_ok(a)) {   // weight-resident stem (bf16 x 3 planes), one workgroup per CU
            ConvPlan sq; sq.stem = 256; cands.push_back(sq);
        }
        if (n->split_precision && lateral_ok(a, groups)) {   // pixel-resident lateral product (bf16 x 3 planes)
            const int tiles = groups * (a.Cout / 32);
            for (int parts = 1; parts <= tiles; parts *= 2)
                if (tiles % parts == 0) { ConvPlan lq; lq.lat = parts; cands.push_back(lq); }
        }
        if (a.wino_w[0] && !a.p[0].up) {
            ConvPlan wq;
            wq.wino = 1; cands.push_back(wq);
            wq.wino = 2; cands.push_back(wq);
            wq.wino = 3; cands.push_back(wq);
            if (a.zeros) { wq.wino = 4; cands.push_back(wq); }
            if (n->split_precision && a.zeros) { wq.wino = 5; cands.push_back(wq); }      // split-precision products, 8 waves
            if (n->split_precision && a.Cout % 128 == 0) { wq.wino = 6; cands.push_back(wq); }      // ... 128 channels per workgroup, 4 waves
            if (n->split_precision) { wq.wino = 7; cands.push_back(wq); }      // ... 64 channels, four waves of 512 registers, weights direct
            // ... on two fp16 pieces (range-limited: fpc.h): all four piece products, or (level 3, Cin a multiple of 16) three of them over
            // pairs of K-steps.  Where the second form is allowed it REPLACES the first as a candidate: one launch timed from a cold
            // clock ranks them by their entry costs, the forward in steady state by their power (all sites on form 9 against all on
            // form 8: 14.05 / 14.58 and 14.21 / 15.04 ms on two boxes, while per-site timing picked form 9 for 2 of 33 sites)
            if (n->split_precision >= 3 && a.Cin % 16 == 0) { wq.wino = 9; cands.push_back(wq); }
            else if (n->split_precision >= 2) { wq.wino = 8; cands.push_back(wq); }
        }
        for (const ConvPlan& q : cands) {
            if (splitk_floats_for(q, groups, a.B, a.Npad) > cap) continue;
            int rc = launch_conv_plan(a, q, groups, s);     // warm-up (also validates the launch)
            if (rc == FPC_EINVAL) continue;                 // a candidate whose launcher refuses this site (its own preconditions) is skipped
            if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
            float ms = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                float t = 0.f;
                bool timed = hipEventRecord(e0, s) == hipSuccess && launch_conv_plan(a, q, groups, s) == FPC_OK &&
                             hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                             hipEventElapsedTime(&t, e0, e1) == hipSuccess;
                if (timed && t < ms) ms = t;             // a candidate that cannot be timed keeps ms = 1e30: never chosen
            }
            // objective: latency, or (throughput mode) latency x the share of the chip the launch occupies —
            // with several frames in flight a launch that leaves CUs free lets another stream's kernels run
            float score = ms;
            if (n->tune_mode >= 1) {
                double nblk = q.stem ? 512.0      // (a persistent 512-thread, 86 KB workgroup per CU: the whole chip, whatever its grid)
                              : q.lat ? (double)cdiv(a.Ho * a.Wo, 128) * a.B * q.lat
                              : q.wino ? (double)cdiv(cdiv(a.Wo, 2), 8) * cdiv(cdiv(a.Ho, 2), (q.wino == 2 || q.wino == 4 || q.wino == 5 || q.wino == 7 || q.wino == 8 || q.wino == 9) ? 8 : 4) * a.B * (a.Cout / (q.wino == 6 ? 128 : 64)) * groups
                                     : (double)q.mtiles * q.ntiles * q.nsplit * a.B * groups;
                double slots = 256.0 * ((q.wino == 2 || q.wino == 4 || q.wino == 5 || q.wino == 6 || q.wino == 7 || q.wino == 8 || q.wino == 9) ? 1.0 : 2.0);
                double share = nblk / slots;
                if (share > 1.0) share = 1.0;
                if (share < 0.125) share = 0.125;
                score = ms * (float)(n->tune_mode == 2 ? share : sqrt(share));      // 2: latency x share = the launch's CU-time
            }
            if (score < best_ms) { best_ms = score; best = q; }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        n->cplan[ci] = best;
    }
    ConvPlan p = n ? n->cplan[ci] : ConvPlan{a.bm, a.bn, a.nsplit, a.mtiles, a.ntiles, 0, a.bf3, a.fused};
    return launch_conv_plan(a, p, groups, s);
}

}  // namespace

#define FPC_TRY(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// Everything between the NCHW -> NHWC4 conversion of the caller's image and the final upsample / class
// compression into the caller's tensors: ~57 launches that touch only the plan's workspace and parameters, i.e.
// identical every frame — the part that can be replayed as a HIP graph.
static int forward_middle(fpc_net* n, hipStream_t s) {
    float* ws = n->ws;
    const int B = n->B, H = n->H, W = n->W;
    ConvArgs a;
    auto nhwc = [&](const Act& t, long long& sb, long long& sh, long long& sw, long long& sc) {
        sc = 1; sw = t.C; sh = (long long)t.W * t.C; sb = (long long)t.H * sh;
    };
    {
        const PackedConv& c = n->convs[n->c_stem];
        // The 7x7/2 stem as a 7x1 convolution over "pixels" of 8 x 4 channels: one K-step = one kernel row = the 8
        // consecutive 16-byte pixels starting at wi0, which are 128 contiguous bytes of the NHWC4 image — the fast
        // loader's case (scalar tap walk, two vector instructions per row and step) instead of the per-lane tap
        // decode of MODE 2 (3500 cycles per K-step of 16 MFMAs, measured).  The eighth tap has zero weights.
        fill_conv_args(n, a, c, n->cplan[n->c_stem], H, W, n->a_stem.H, n->a_stem.W, (long long)4 * H * W, (long long)4 * W, 4,
                       1, true, 0);
        a.Cin = 8 * c.Cinp; a.Kw = 1; a.K = c.K; a.lanepx = 1;
        a.p[0] = ConvPtrs{ws + n->a_img4.off, ws + c.w_off, ws + n->a_stem.off, ws + c.scale_off, ws + c.shift_off, nullptr, nullptr, nullptr};
        FPC_TRY(run_conv(n, a, 1, n->c_stem, s));
    }
    FPC_TRY(launch_maxpool3x3s2(ws + n->a_stem.off, ws + n->a_pool.off, B, n->a_stem.H, n->a_stem.W, 64, n->a_pool.H,
                                n->a_pool.W, s));
    // encoder stages
    Act cur = n->a_pool;
    Act feat[4];
    for (int L = 0; L < 4; ++L) {
        for (size_t bi = 0; bi < n->blocks[L].size(); ++bi) {
            const fpc_net::Block& blk = n->blocks[L][bi];
            const Act& T = n->a_blk_t[L][bi];
            const Act& Y = n->a_blk_y[L][bi];
            long long sb, sh, sw, sc;
            nhwc(cur, sb, sh, sw, sc);
            const PackedConv& c1 = n->convs[blk.conv1];
            fill_conv_args(n, a, c1, n->cplan[blk.conv1], cur.H, cur.W, T.H, T.W, sb, sh, sw, sc, true, 0);
            a.p[0] = ConvPtrs{ws + cur.off, ws + c1.w_off, ws + T.off, ws + c1.scale_off, ws + c1.shift_off, nullptr, nullptr, nullptr};
            if (c1.wino_ok) a.wino_w[0] = ws + c1.wino_off;
            FPC_TRY(run_conv(n, a, 1, blk.conv1, s));
            const float* res = ws + cur.off;
            if (blk.ds >= 0) {
                const PackedConv& cd = n->convs[blk.ds];
                const Act& D = n->a_blk_d[L][bi];
                fill_conv_args(n, a, cd, n->cplan[blk.ds], cur.H, cur.W, D.H, D.W, sb, sh, sw, sc, false, 0);
                a.p[0] = ConvPtrs{ws + cur.off, ws + cd.w_off, ws + D.off, ws + cd.scale_off, ws + cd.shift_off, nullptr, nullptr, nullptr};
                FPC_TRY(run_conv(n, a, 1, blk.ds, s));
                res = ws + D.off;
            }
            nhwc(T, sb, sh, sw, sc);
            const PackedConv& c2 = n->convs[blk.conv2];
            fill_conv_args(n, a, c2, n->cplan[blk.conv2], T.H, T.W, Y.H, Y.W, sb, sh, sw, sc, true, 0);
            a.p[0] = ConvPtrs{ws + T.off, ws + c2.w_off, ws + Y.off, ws + c2.scale_off, ws + c2.shift_off, res, nullptr, nullptr};
            if (c2.wino_ok) a.wino_w[0] = ws + c2.wino_off;
            FPC_TRY(run_conv(n, a, 1, blk.conv2, s));
            cur = Y;
        }
        feat[L] = cur;
    }

    // ---- four decoders, grouped
    // laterals: p5 = conv(c5); p4 = up2_nearest(p5) + conv(c4); p3; p2
    for (int i = 0; i < 4; ++i) {
        const Act& src = feat[3 - i];
        long long sb, sh, sw, sc;
        nhwc(src, sb, sh, sw, sc);
        int ci0 = n->dec[0].lat[i];
        fill_conv_args(n, a, n->convs[ci0], n->cplan[ci0], src.H, src.W, src.H, src.W, sb, sh, sw, sc, false, 0);
        for (int d = 0; d < 4; ++d) {
            const PackedConv& c = n->convs[n->dec[d].lat[i]];
            a.p[d] = ConvPtrs{ws + src.off, ws + c.w_off, ws + n->a_p[d][i].off, nullptr, n->pptr[c.p_bias], nullptr,
                              i > 0 ? ws + n->a_p[d][i - 1].off : nullptr, nullptr};
        }
        FPC_TRY(run_conv(n, a, 4, ci0, s));
    }
    // segmentation blocks
    auto seg_conv = [&](int si, int which) -> int {
        // input of decoder d: which < 0 -> a_p[d][-which-1], else a_up[d][which]
        int ci0 = n->dec[0].seg[si];
        const Act& in0 = which < 0 ? n->a_p[0][-which - 1] : n->a_up[0][which];
        long long sb, sh, sw, sc;
        nhwc(in0, sb, sh, sw, sc);
        const Act& o0 = n->a_seg[0][si];
        fill_conv_args(n, a, n->convs[ci0], n->cplan[ci0], in0.H, in0.W, o0.H, o0.W, sb, sh, sw, sc, false, 0);
        for (int d = 0; d < 4; ++d) {
            const PackedConv& c = n->convs[n->dec[d].seg[si]];
            const Act& in = which < 0 ? n->a_p[d][-which - 1] : n->a_up[d][which];
            a.p[d] = ConvPtrs{ws + in.off, ws + c.w_off, ws + n->a_seg[d][si].off, nullptr, nullptr, nullptr, nullptr,
                              ws + n->gn_part_off[d][si]};
            if (c.wino_ok) a.wino_w[d] = ws + c.wino_off;
        }
        return run_conv(n, a, 4, ci0, s);
    };
    // GroupNorm statistics of up to kMaxGnSites finished sites -> per (image, channel) affines, one launch
    auto gn_finish = [&](std::initializer_list<int> sites) -> int {
        GnFinArgs g;
        memset(&g, 0, sizeof(g));
        int k = 0;
        for (int si : sites) {
            const Act& o0 = n->a_seg[0][si];
            for (int d = 0; d < 4; ++d) {
                g.gn_part[k * kMaxGroup + d] = ws + n->gn_part_off[d][si];
                g.gamma[k * kMaxGroup + d] = n->pptr[n->dec[d].p_gn[si]];
                g.beta[k * kMaxGroup + d] = n->pptr[n->dec[d].p_gn[si] + 1];
                g.affine[k * kMaxGroup + d] = ws + n->gn_aff_off[d][si];
            }
            g.P[k] = plan_gn_rows(n->cplan[n->dec[0].seg[si]], o0.H, o0.W);
            g.count[k] = (long long)o0.H * o0.W * 4;
            ++k;
        }
        g.B = B; g.C = 128; g.groups = 32; g.sites = k; g.eps = 1e-5f;
        return launch_gn_finalize(g, 4, s);
    };
    // GN + ReLU + x2 upsample of up to kMaxUpJobs sites (si -> a_up[ui]), one launch
    auto gn_up = [&](std::initializer_list<std::pair<int, int>> jobs) -> int {
        GnUpArgs u;
        memset(&u, 0, sizeof(u));
        int k = 0;
        for (const auto& job : jobs) {
            const int si = job.first, ui = job.second;
            for (int d = 0; d < 4; ++d) {
                u.in[k * kMaxGroup + d] = ws + n->a_seg[d][si].off;
                u.affine[k * kMaxGroup + d] = ws + n->gn_aff_off[d][si];
                u.out[k * kMaxGroup + d] = ws + n->a_up[d][ui].off;
            }
            u.h[k] = n->a_seg[0][si].H; u.w[k] = n->a_seg[0][si].W;
            ++k;
        }
        u.B = B; u.C = 128; u.jobs = k;
        return launch_gn_relu_up2(u, 4, s);
    };
    // the statistics of several sites are finalized together and both first-level upsamples share a launch:
    // 3 + 2 small launches per frame instead of 7 + 3 (each ~4.7 us of latency at batch 1)
    FPC_TRY(seg_conv(0, -1));   // s5.0 on p5
    FPC_TRY(seg_conv(3, -2));   // s4.0 on p4
    FPC_TRY(seg_conv(5, -3));   // s3.0 on p3
    FPC_TRY(seg_conv(6, -4));   // s2.0 on p2
    FPC_TRY(gn_finish({0, 3, 5, 6}));
    FPC_TRY(gn_up({{0, 0}, {3, 2}}));
    FPC_TRY(seg_conv(1, 0));    // s5.1 on up(s5.0)
    FPC_TRY(seg_conv(4, 2));    // s4.1 on up(s4.0)
    FPC_TRY(gn_finish({1, 4}));
    FPC_TRY(gn_up({{1, 1}}));
    FPC_TRY(seg_conv(2, 1));    // s5.2 on up(s5.1)
    FPC_TRY(gn_finish({2}));

    // merge + head
    const int lo[3] = {2, 4, 5};     // s5.2, s4.1, s3.0 (sum order of the reference: p5-, p4-, p3-, p2-branch)
    const int split = n->merge_split;      // resolved once in fpc_net_create (FPC_MERGE_SPLIT): graph and plain runs agree, no getenv per forward
    if (split) {
        // two passes (merge_split.hip): the head of (r5 + r4 + r3) at the branches' resolution, then the head of r2 + bias + its x2 upsample
        HeadPartArgs hl, hh;
        memset(&hl, 0, sizeof(hl));
        memset(&hh, 0, sizeof(hh));
        for (int d = 0; d < 4; ++d) {
            for (int k = 0; k < 3; ++k) {
                hl.t[d][k] = ws + n->a_seg[d][lo[k]].off;
                hl.aff[d][k] = ws + n->gn_aff_off[d][lo[k]];
            }
            hh.t[d][0] = ws + n->a_seg[d][6].off;
            hh.aff[d][0] = ws + n->gn_aff_off[d][6];
            hl.hw[d] = hh.hw[d] = n->pptr[n->dec[d].p_head_w];
            hl.hb[d] = hh.hb[d] = n->pptr[n->dec[d].p_head_b];
            hl.out[d] = ws + n->a_lsum[d].off;
            hh.lsum[d] = ws + n->a_lsum[d].off;
            hh.out[d] = ws + n->a_low[d].off;
            hl.ch[d] = hh.ch[d] = n->dec[d].head_ch;
            hl.chp[d] = hh.chp[d] = n->dec[d].head_chp;
        }
        hl.B = hh.B = B; hl.C = hh.C = 128;
        hl.H = n->a_seg[0][2].H; hl.W = n->a_seg[0][2].W; hl.hl = hl.H; hl.wl = hl.W;
        hh.H = n->a_seg[0][6].H; hh.W = n->a_seg[0][6].W; hh.hl = hl.H; hh.wl = hl.W;
        FPC_TRY(launch_head_part(hl, false, 4, s));
        FPC_TRY(launch_head_part(hh, true, 4, s));
    } else {
        MergeHeadArgs m;
        memset(&m, 0, sizeof(m));
        for (int d = 0; d < 4; ++d) {
            for (int k = 0; k < 3; ++k) {
                m.t_lo[d][k] = ws + n->a_seg[d][lo[k]].off;
                m.a_lo[d][k] = ws + n->gn_aff_off[d][lo[k]];
            }
            m.t_hi[d] = ws + n->a_seg[d][6].off;
            m.a_hi[d] = ws + n->gn_aff_off[d][6];
            m.hw[d] = n->pptr[n->dec[d].p_head_w];
            m.hb[d] = n->pptr[n->dec[d].p_head_b];
            m.out[d] = ws + n->a_low[d].off;
            m.ch[d] = n->dec[d].head_ch;
            m.chp[d] = n->dec[d].head_chp;
        }
        m.B = B; m.h = n->a_seg[0][2].H; m.w = n->a_seg[0][2].W; m.C = 128;
        FPC_TRY(launch_merge_head(m, 4, s));
    }
    return FPC_OK;
}

extern "C" int fpc_net_forward(fpc_net_t* n, const float* x, float* logits_mask, float* logits_quat,
                               float* logits_scales, float* logits_xy, float* logits_z, int64_t* cat_mask, float* cq,
                               float* cs, float* cxy, float* cz, fpc_stream_t stream) {
    return fpc_net_forward_bits(n, x, logits_mask, logits_quat, logits_scales, logits_xy, logits_z, cat_mask, cq, cs, cxy, cz, nullptr, stream);
}

extern "C" int fpc_net_forward_bits(fpc_net_t* n, const float* x, float* logits_mask, float* logits_quat,
                                    float* logits_scales, float* logits_xy, float* logits_z, int64_t* cat_mask, float* cq,
                                    float* cs, float* cxy, float* cz, uint64_t* fg_bits, fpc_stream_t stream) {
    if (!n || !n->loaded || !x || !cat_mask || !cq || !cs || !cxy || !cz) return FPC_EINVAL;
    if (fg_bits && (n->W % 64 != 0 || ((uintptr_t)fg_bits & 7))) return FPC_EINVAL;
    bool any = logits_mask || logits_quat || logits_scales || logits_xy || logits_z;
    bool all = logits_mask && logits_quat && logits_scales && logits_xy && logits_z;
    if (any && !all) return FPC_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* ws = n->ws;
    const int B = n->B, H = n->H, W = n->W;

    // stem: image -> NHWC4 (16-byte pixels), 7x7/2 with BN + ReLU in the epilogue
    FPC_TRY(launch_nchw3_to_nhwc4(x, ws + n->a_img4.off, B, H * W, s));
    // graph replay needs a capturable stream: not the null (legacy default) stream
    if (n->use_graph && !n->tuning && s != nullptr && !n->graph_exec) {
        // first such frame after tuning: record the launches instead of running them
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = forward_middle(n, s);
            hipError_t e = hipStreamEndCapture(s, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e == hipSuccess && g) e = hipGraphInstantiate(&n->graph_exec, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess) n->graph_exec = nullptr;
        }
        if (!n->graph_exec) { (void)hipGetLastError(); n->use_graph = 0; }      // not capturable here: plain launches from now on
    }
    if (n->use_graph && !n->tuning && s != nullptr && n->graph_exec) {
        if (hipGraphLaunch(n->graph_exec, s) != hipSuccess) return FPC_ELAUNCH;
    } else {
        FPC_TRY(forward_middle(n, s));
    }
    {
        Up4Args u;
        memset(&u, 0, sizeof(u));
        u.lm = ws + n->a_low[0].off; u.lq = ws + n->a_low[1].off; u.lt = ws + n->a_low[2].off; u.ls = ws + n->a_low[3].off;
        u.pm = n->dec[0].head_chp; u.pq = n->dec[1].head_chp; u.pt = n->dec[2].head_chp; u.ps = n->dec[3].head_chp;
        u.o_mask = logits_mask; u.o_quat = logits_quat; u.o_scales = logits_scales; u.o_xy = logits_xy; u.o_z = logits_z;
        u.cat_mask = (long long*)cat_mask; u.cq = cq; u.cs = cs; u.cxy = cxy; u.cz = cz;
        u.B = B; u.hl = n->a_low[0].H; u.wl = n->a_low[0].W; u.H = H; u.W = W; u.C = n->classes;
        u.fg_bits = reinterpret_cast<unsigned long long*>(fg_bits); u.fg_stride = (size_t)((H * W + 4095) / 4096) * 64;
        if (u.hl * 4 != H || u.wl * 4 != W) return FPC_EINVAL;
        FPC_TRY(launch_up4_compress(u, s));
    }
    if (n->tuning) { n->tuning = false; n->tuned = true; }
    return FPC_OK;
}

// The NEXT fpc_net_forward times every candidate tiling of every convolution site on the device
// (it synchronises the stream; not capturable) and keeps the fastest; later forwards reuse the plans.
extern "C" int fpc_net_autotune_next(fpc_net_t* n, int mode) {
    if (!n || !n->loaded || mode < 0 || mode > 2) return FPC_EINVAL;
    n->tuning = true;
    n->tune_mode = mode;
    if (n->graph_exec) { (void)hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; }      // tilings may change
    return FPC_OK;
}

// Chosen tiling of convolution site `i` (0 <= i < fpc_net_conv_count): out5 = bm, bn, nsplit, Cout, K.
extern "C" int fpc_net_conv_count(const fpc_net_t* n) { return n ? (int)n->convs.size() : 0; }
extern "C" int fpc_net_conv_plan(const fpc_net_t* n, int i, int* out5) {
    if (!n || !out5 || i < 0 || i >= (int)n->convs.size()) return FPC_EINVAL;
    out5[0] = n->cplan[i].bm; out5[1] = n->cplan[i].bn; out5[2] = n->cplan[i].wino ? -n->cplan[i].wino : n->cplan[i].nsplit;
    if (n->cplan[i].lat) { out5[0] = 128; out5[1] = 32; out5[2] = 2000 + n->cplan[i].lat; }      // k_lateral1x1 (fpc_conv2d's hook value)
    if (n->cplan[i].stem) { out5[0] = 64; out5[1] = 64; out5[2] = 3000; }                        // k_stem7x7
    out5[3] = n->convs[i].Cout; out5[4] = n->convs[i].K;
    return FPC_OK;
}

// Every 3x3 / stride-1 site that has Winograd images -> Winograd form `form` (1..8, fpc_conv2d's -form; 6 only where Cout % 128 == 0,
// other sites keep their plan): tests run the whole network on ONE form (e.g. 8: every eligible product on fp16 x 2 pieces) and
// hold it to the float64 bars.  Returns the number of sites changed, or a negative code.  Drops the recorded graph.
extern "C" int fpc_net_force_winograd(fpc_net_t* n, int form) {
    if (!n || form < 1 || form > 9) return FPC_EINVAL;
    int changed = 0;
    for (size_t i = 0; i < n->convs.size(); ++i) {
        const PackedConv& c = n->convs[i];
        if (!c.wino_ok || !n->c_groups[i] || (form == 6 && c.Cout % 128 != 0) || (form == 9 && c.Cin % 16 != 0)) continue;
        ConvPlan q = n->cplan[i];
        q.wino = form; q.lat = 0; q.stem = 0;
        n->cplan[i] = q;
        ++changed;
    }
    if (n->graph_exec) { (void)hipGraphExecDestroy(n->graph_exec); n->graph_exec = nullptr; }
    return changed;
}

// Plans of `src` -> `dst` (same encoder, classes, H, W; batch sizes may differ): runs a small batch on the tilings, split-K
// factors and kernel forms a larger one was autotuned to (tests: the headline configuration's kernels against float64 on two
// frames).  A plan whose split-K partials do not fit dst's workspace keeps dst's own.  Drops dst's recorded graph.
extern "C" int fpc_net_copy_plans(fpc_net_t* dst, const fpc_net_t* src) {
    if (!dst || !src || dst->convs.size() != src->convs.size() || dst->H != src->H || dst->W != src->W) return FPC_EINVAL;
    for (size_t i = 0; i < dst->convs.size(); ++i) {
        const PackedConv &a = dst->convs[i], &b = src->convs[i];
        if (a.Cin != b.Cin || a.Cout != b.Cout || a.Kh != b.Kh || a.Kw != b.Kw || a.stride != b.stride || a.Npad != b.Npad ||
            a.Kpad != b.Kpad || dst->c_groups[i] != src->c_groups[i])
            return FPC_EINVAL;
    }
    for (size_t i = 0; i < dst->convs.size(); ++i) {
        const ConvPlan& q = src->cplan[i];
        if (splitk_floats_for(q, dst->c_groups[i] ? dst->c_groups[i] : 1, dst->B, dst->convs[i].Npad) > dst->splitk_floats) continue;
        if (q.nsplit > 1 && q.fused && !can_fuse(q, dst->c_groups[i] ? dst->c_groups[i] : 1, dst->B)) continue;
        dst->cplan[i] = q;
    }
    dst->tuned = true;
    if (dst->graph_exec) { (void)hipGraphExecDestroy(dst->graph_exec); dst->graph_exec = nullptr; }
    return FPC_OK;
}

// FLOP of one forward over the whole batch: out3[0] = 2 x MACs of the direct convolutions (the algorithmic count the
// reference's cuDNN path would execute), out3[1] = multiply-adds the CURRENT plans execute (a Winograd F(2x2,3x3)
// site does 16 instead of 36 per 2x2 output tile: direct / 2.25), out3[2] = share of out3[0] on Winograd sites.
extern "C" int fpc_net_flops(const fpc_net_t* n, double* out3) {
    if (!n || !out3) return FPC_EINVAL;
    double direct = 0.0, executed = 0.0, wino = 0.0;
    for (size_t i = 0; i < n->convs.size(); ++i) {
        if (!n->c_groups[i]) continue;
        const PackedConv& c = n->convs[i];
        const double f = 2.0 * n->B * (double)n->c_howo[i] * c.Cout * c.Cin * c.Kh * c.Kw * n->c_groups[i];
        direct += f;
        if (n->cplan[i].wino) { executed += f / 2.25; wino += f; } else executed += f;
    }
    for (int d = 0; d < 4; ++d)      // the 1x1 heads run inside k_merge_head
        direct += 2.0 * n->B * (double)n->a_low[d].H * n->a_low[d].W * 128.0 * n->dec[d].head_ch,
        executed += 2.0 * n->B * (double)n->a_low[d].H * n->a_low[d].W * 128.0 * n->dec[d].head_ch;
    out3[0] = direct; out3[1] = executed; out3[2] = direct > 0.0 ? wino / direct : 0.0;
    return FPC_OK;
}

// Debug / test access to the engine's intermediate activations (NHWC f32 inside the workspace).
// name: "stem", "pool", "c2".."c5", "d<k>.p5".."d<k>.p2", "d<k>.seg<i>" (pre-GroupNorm), "d<k>.low".
extern "C" int fpc_net_tensor(const fpc_net_t* n, const char* name, const float** ptr, int* H, int* W, int* C) {
    if (!n || !n->ws || !name || !ptr || !H || !W || !C) return FPC_EINVAL;
    Act t;
    bool ok = false;
    if (!strcmp(name, "stem")) { t = n->a_stem; ok = true; }
    else if (!strcmp(name, "pool")) { t = n->a_pool; ok = true; }
    else if (name[0] == 'c' && name[1] >= '2' && name[1] <= '5' && !name[2]) { t = n->a_blk_y[name[1] - '2'].back(); ok = true; }
    else if (name[0] == 'd' && name[1] >= '0' && name[1] <= '3' && name[2] == '.') {
        int d = name[1] - '0';
        const char* r = name + 3;
        if (r[0] == 'p' && r[1] >= '2' && r[1] <= '5' && !r[2]) { t = n->a_p[d]['5' - r[1]]; ok = true; }
        else if (!strncmp(r, "seg", 3) && r[3] >= '0' && r[3] <= '6' && !r[4]) { t = n->a_seg[d][r[3] - '0']; ok = true; }
        else if (!strcmp(r, "low")) { t = n->a_low[d]; ok = true; }
    }
    if (!ok) return FPC_EINVAL;
    *ptr = n->ws + t.off; *H = t.H; *W = t.W; *C = t.C;
    return FPC_OK;
}

// ---- stand-alone convolution (unit tests / micro-benchmarks of k_conv_igemm) -----------------
extern "C" size_t fpc_conv2d_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw) {
    int K = Cin * Kh * Kw, Kpad = cdiv(K, kConvBK) * kConvBK, Npad = cdiv(Cout, kConvNAlign) * kConvNAlign;
    size_t packed = conv_packed_floats(Npad, Kpad);
    size_t splitk = (size_t)32 * B * (cdiv(Ho * Wo, 128) * 128) * Npad;
    size_t wino = (size_t)96 * Cout * Cin + 128;      // f32 + split-precision (two layouts) Winograd images + a zero page for the all-DMA form
    return (packed + splitk + wino + kConvTickets) * sizeof(float);
}

extern "C" int fpc_conv2d_plan(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw, int bm, int bn, int nsplit,
                               int* out4) {
    if (!out4) return FPC_EINVAL;
    int Kpad = cdiv(Cin * Kh * Kw, kConvBK) * kConvBK;
    if (nsplit >= 2000) nsplit = 1;          // k_lateral1x1 / k_stem7x7: no split-K, no GroupNorm rows
    if (nsplit >= 1000) nsplit -= 1000;      // fpc_conv2d's split-precision / two-launch hooks do not change the tiling
    if (nsplit >= 100) nsplit -= 100;
    ConvPlan p = plan_conv(Ho * Wo, B, Cout, Kpad / kConvBK, 1, bm, bn, nsplit);
    if (nsplit <= -1 && nsplit >= -9) { p.wino = -nsplit; p.nsplit = nsplit; }
    out4[0] = p.bm; out4[1] = p.bn; out4[2] = p.nsplit; out4[3] = plan_gn_rows(p, Ho, Wo);
    return FPC_OK;
}

namespace {
// the hooks folded into fpc_conv2d's `nsplit` argument
struct Conv2dRequest { int nsplit; bool bf3, two_launch, wino; int lat; bool stem; };
Conv2dRequest conv2d_request(int nsplit) {
    Conv2dRequest r{nsplit, false, false, false, 0, false};
    if (r.nsplit == 3000) { r.stem = true; r.bf3 = true; r.nsplit = 1; return r; }                   // 3000 = k_stem7x7 (stem.hip): NHWC4 input
    if (r.nsplit >= 2000) { r.lat = r.nsplit - 2000; r.bf3 = true; r.nsplit = 1; return r; }      // 2000 + parts = k_lateral1x1 (lateral.hip)
    if (r.nsplit >= 1000) { r.bf3 = true; r.nsplit -= 1000; }          // 1000 + split = split-precision matrix products
    if (r.nsplit >= 100) { r.two_launch = true; r.nsplit -= 100; }      // 100 + split = split-K summed by k_conv_splitk_epilogue
    // -1: 4 waves, -2: 8 waves, -3: wave-private, -4: all-DMA 3-stage, -5: 8 waves split precision, -6: split precision, 128 channels per workgroup
    // -7: split precision, 64 channels, four waves of 512 registers (the -5 image)
    // -8: the -7 form on two fp16 pieces per operand (its own image)
    // -9: three of the four products of -8, over pairs of K-steps (its own image; Cin a multiple of 16)
    r.wino = r.nsplit <= -1 && r.nsplit >= -9;
    return r;
}
// workspace of ONE fpc_conv2d call (floats): [packed weights | split-K partials of this plan | Winograd images + zero page |
// arrival counters]; regions a plan does not use are empty
struct Conv2dLayout { size_t packed, splitk, wino, tickets, total; };
Conv2dLayout conv2d_layout(int B, int Cin, int Cout, int Kh, int Kw, const ConvPlan& p, const Conv2dRequest& r) {
    const int K = Cin * Kh * Kw, Kpad = cdiv(K, kConvBK) * kConvBK, Npad = cdiv(Cout, kConvNAlign) * kConvNAlign;
    Conv2dLayout L;
    L.packed = r.wino ? 0 : conv_packed_floats(Npad, Kpad);
    L.splitk = r.wino ? 0 : (splitk_floats_for(p, 1, B, Npad) + 63) / 64 * 64;
    L.wino = r.wino ? (size_t)(r.nsplit == -9 ? (Cout % 128 == 0 ? 96 : 72) : r.nsplit == -8 ? (Cout % 128 == 0 ? 80 : 56) : r.nsplit == -6 ? 64 : (r.nsplit == -5 || r.nsplit == -7) ? 40 : 16) * Cout * Cin + 128 : 0;
    L.tickets = (!r.wino && p.fused && p.nsplit > 1) ? kConvTickets : 0;
    L.total = L.packed + L.splitk + L.wino + L.tickets;
    return L;
}
ConvPlan conv2d_plan_for(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw, int bm, int bn, const Conv2dRequest& r) {
    const int Kpad = cdiv(Cin * Kh * Kw, kConvBK) * kConvBK;
    ConvPlan p = plan_conv(Ho * Wo, B, Cout, Kpad / kConvBK, 1, r.wino ? 0 : bm, bn, r.wino ? 1 : r.nsplit);
    p.bf3 = r.bf3 ? 1 : 0;
    if (r.two_launch) p.fused = 0;
    p.lat = r.lat;
    p.stem = r.stem ? 256 : 0;
    return p;
}
}  // namespace

// Exact workspace of fpc_conv2d for ONE request (same bm / bn / nsplit): at most fpc_conv2d_workspace_bytes, usually far
// less (that bound reserves 32 split-K slices of the whole output).  fpc_conv2d accepts either.
extern "C" size_t fpc_conv2d_workspace_bytes_for(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw, int bm, int bn,
                                                 int nsplit) {
    if (B < 1 || Ho < 1 || Wo < 1 || Cin < 1 || Cout < 1 || Kh < 1 || Kw < 1) return 0;
    const Conv2dRequest r = conv2d_request(nsplit);
    const ConvPlan p = conv2d_plan_for(B, Ho, Wo, Cin, Cout, Kh, Kw, bm, bn, r);
    return std::max<size_t>(conv2d_layout(B, Cin, Cout, Kh, Kw, p, r).total * sizeof(float), 256);
}

extern "C" int fpc_conv2d(const float* in, int64_t sb, int64_t sh, int64_t sw, int64_t sc, const float* w_oihw,
                          const float* scale, const float* shift, const float* res, const float* up, float* out,
                          float* gn_part, int B, int Hi, int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad,
                          int relu, int bm, int bn, int nsplit, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (!in || !w_oihw || !out || !ws || B < 1 || Kh != Kw) return FPC_EINVAL;
    int Ho = conv_out(Hi, Kh, stride, pad), Wo = conv_out(Wi, Kw, stride, pad);
    if (Ho < 1 || Wo < 1) return FPC_EINVAL;
    const Conv2dRequest rq = conv2d_request(nsplit);
    nsplit = rq.nsplit;
    const bool wino = rq.wino;
    ConvPlan p = conv2d_plan_for(B, Ho, Wo, Cin, Cout, Kh, Kw, bm, bn, rq);
    const Conv2dLayout lay = conv2d_layout(B, Cin, Cout, Kh, Kw, p, rq);
    if (ws_bytes < lay.total * sizeof(float) || ((uintptr_t)ws & 255)) return FPC_EWORKSPACE;
    PackedConv c;
    c.Cin = c.Cinp = Cin; c.Cout = Cout; c.Kh = Kh; c.Kw = Kw; c.stride = stride; c.pad = pad;
    c.K = Cin * Kh * Kw; c.Kpad = cdiv(c.K, kConvBK) * kConvBK; c.Npad = cdiv(Cout, kConvNAlign) * kConvNAlign;
    hipStream_t s = (hipStream_t)stream;
    float* packed = (float*)ws;
    if (rq.stem) {      // the engine's stem layout: NHWC4 input, 8 taps x 4 channels per kernel row (K = 224), bf16 planes only
        if (Cin != 4 || Kh != 7 || stride != 2 || pad != 3 || Cout != 64 || sc != 1 || sw != 4 || sh != (int64_t)4 * Wi ||
            sb != (int64_t)4 * Hi * Wi || res || up || gn_part)
            return FPC_EINVAL;
        c.Kwp = 8; c.K = 4 * 7 * 8; c.Kpad = 224;
        FPC_TRY(launch_pack_weight_bf3(w_oihw, packed, Cout, Cin, Cin, Kh, Kw, c.Kwp, c.Npad, c.Kpad, s));
        fpc_net tmp0;
        tmp0.B = B; tmp0.ws = packed; tmp0.splitk_off = lay.packed;
        ConvArgs a0;
        fill_conv_args(&tmp0, a0, c, p, Hi, Wi, Ho, Wo, sb, sh, sw, sc, relu != 0, 0);
        a0.Cin = 8 * c.Cinp; a0.Kw = 1; a0.K = c.K; a0.lanepx = 1;
        a0.p[0] = ConvPtrs{in, packed, out, scale, shift, nullptr, nullptr, nullptr};
        return launch_conv_plan(a0, p, 1, s);
    }
    // (the split-precision forms — bf16 x 3 tiles, k_lateral1x1 — read only the planes behind the f32 image: it is not packed for them)
    static_assert(FPC_IGEMM_DMA_B, "the split-precision direct form stages its B rows from the bf16 planes by LDS-DMA; a build that loads "
                                   "them from the f32 image must pack that image here as well");
    if (!wino && !p.bf3) FPC_TRY(launch_pack_weight(w_oihw, packed, Cout, Cin, Cin, Kh, Kw, Kw, c.Npad, c.Kpad, s));
    if (!wino && p.bf3) FPC_TRY(launch_pack_weight_bf3(w_oihw, packed, Cout, Cin, Cin, Kh, Kw, Kw, c.Npad, c.Kpad, s));
    int mode = (sc == 1 && Cin % kConvBK == 0 && Kh * Kw <= 32 && ((int64_t)Hi + 2 * pad) * sh * 4 < ((int64_t)1 << 31)) ? 0
               : (sc == 1 && Cin % 4 == 0 && sw % 4 == 0 && sh % 4 == 0 && sb % 4 == 0 && ((uintptr_t)in & 15) == 0) ? 2 : 1;
    fpc_net tmp;
    tmp.B = B;
    tmp.ws = packed;
    tmp.splitk_off = lay.packed;
    ConvArgs a;
    fill_conv_args(&tmp, a, c, p, Hi, Wi, Ho, Wo, sb, sh, sw, sc, relu != 0, mode);
    a.p[0] = ConvPtrs{in, packed, out, scale, shift, res, up, gn_part};
    a.zeros = nullptr;
    a.tickets = nullptr;
    if (lay.tickets) {   // arrival counters of the fused split-K form, zeroed per call
        float* tk = packed + lay.packed + lay.splitk + lay.wino;
        a.tickets = (int*)tk;
        if (hipMemsetAsync(tk, 0, kConvTickets * sizeof(int), s) != hipSuccess) return FPC_ELAUNCH;
    }
    if (wino && relu == 77) { a.dbg = gn_part; a.p[0].gn_part = nullptr; a.relu = 0; }
#ifdef FPC_STAMP_IGEMM
    if (!wino && relu == 77) { a.dbg = gn_part; a.p[0].gn_part = nullptr; a.relu = 0; }
#endif
    if (wino) {
        if (Kh != 3 || stride != 1 || pad != 1 || Cin % 8 || Cout % 64 || sc != 1 || up || sw != Cin ||
            sh != (int64_t)Wi * Cin || sb != (int64_t)Hi * Wi * Cin)
            return FPC_EINVAL;
        float* wp = packed + lay.packed + lay.splitk;
        // (the split-precision form reads only its own image: the f32 image is not packed for it — 78 launches of a training step)
        if (nsplit == -6 && Cout % 128) return FPC_EINVAL;
        if (nsplit > -5) FPC_TRY(launch_wino_pack(w_oihw, wp, Cout, Cin, s));
        if (nsplit == -5 || nsplit == -7) FPC_TRY(launch_wino_pack_bf3(w_oihw, wp + (size_t)16 * Cout * Cin, Cout, Cin, s));
        if (nsplit == -6) FPC_TRY(launch_wino_pack_c128(w_oihw, wp + (size_t)40 * Cout * Cin, Cout, Cin, s));
        if (nsplit == -9) FPC_TRY(launch_wino_pack_h3(w_oihw, wp + (size_t)(Cout % 128 == 0 ? 80 : 56) * Cout * Cin + 8, Cout, Cin, s));
        if (nsplit == -8) FPC_TRY(launch_wino_pack_h2(w_oihw, wp + (size_t)(Cout % 128 == 0 ? 64 : 40) * Cout * Cin, Cout, Cin, s));
        a.wino_w[0] = wp;
        a.zeros = zero_page();       // (the workspace's last 64 floats stay reserved for it: fpc_conv2d_workspace_bytes is unchanged)
        if (!a.zeros) {
            float* zp = wp + lay.wino - 64;
            if (hipMemsetAsync(zp, 0, 64 * sizeof(float), s) != hipSuccess) return FPC_ELAUNCH;
            a.zeros = zp;
        }
        p.wino = -nsplit;
        return launch_conv_plan(a, p, 1, s);
    }
    a.bf3 = (p.bf3 && mode == 0) ? 1 : 0;
    if (p.bf3 && mode != 0) return FPC_EINVAL;
    if (p.lat) return launch_conv_plan(a, p, 1, s);
    return run_conv(nullptr, a, 1, 0, s);
}
