// rebalance.h -- pack-time (host) exact power-of-two rebalancing of a chain of affine layers, for the fp16 x 2 split products' exponent range.
//
// csrc/split2.h represents an operand as two fp16 terms; that is float32-class for |x| in [2^-14, 65504].  Above, the kernels' sticky range flag
// sends the batch to the bf16 x 3 kernels; BELOW (h0 subnormal) the terms lose bits silently: tools/f16x2_probe.sh measured 33 x the float32
// chain's error for a tensor of weights around 1e-5.  A network whose layer i carries tiny weights and whose layer i + 1 undoes it with huge
// ones (per-channel scales folded one way, normalisation constants folded the other) is the same function as the balanced one: between two
// affine layers that only positively homogeneous operations separate (ReLU, depthwise FIR + skip, stride, a linear layer without activation),
//     W_l' = 2^(k_l - k_(l-1)) W_l,   b_l' = 2^(k_l) b_l          (activations of layer l carried at 2^(k_l); k of the last layer = 0)
// changes nothing and is EXACT in float32 (powers of two; no weight here comes near float32's own exponent limits).  rebalance_chain picks the
// k_l from the weights alone, with a wide dead zone so that ordinary checkpoints are left bit-for-bit as they are:
//     u_l = floor(log2 max|W_l|) - k_(l-1)      (the exponent the layer's weights have once the incoming scale is undone)
//     u_l in [REB_LO, REB_HI] -> k_l = 0;   otherwise k_l = REB_MID - u_l   (the weights' exponent moves to the middle of the band).
// The rebalanced weights are what EVERY arithmetic's fragments are packed from (float32 MFMAs and bf16 x 3 too: one network, one set of biases).
// A segment's last layer absorbs the carried scale; segments end wherever a tensor is visible outside the kernels (FIR caches, LSTM inputs,
// scores).  What is still outside the band afterwards is reported to the caller (`min_exp`): pack_host refuses fp16 x 2 for such a blob.
#pragma once
#include <math.h>
#include <stddef.h>
#include <vector>

namespace vadx {

constexpr int REB_LO = -10, REB_HI = 6, REB_MID = -3;       // band of floor(log2 max|W|): [2^-10, 2^7); a layer outside it is moved to [2^-3, 2^-2)
constexpr int REB_REFUSE = -14;                            // a weight tensor whose max is below 2^-14 after rebalancing has no normal fp16 term: no fp16 x 2

struct RebLayer {
    std::vector<float> *w;      // the layer's weights, any layout
    std::vector<float> *b;      // its bias (nullptr: none)
};

inline int reb_exponent(const std::vector<float> &w) {          // floor(log2 max|w|); INT_MIN-like for an all-zero (or non-finite) tensor
    float m = 0.f;
    for (float v : w) { const float a = fabsf(v); if (a > m && a <= 3.0e38f) m = a; }
    if (m == 0.f) return -100000;
    int e;
    frexpf(m, &e);
    return e - 1;
}

// One segment: layers[0] .. layers[n - 1] in evaluation order, only positively homogeneous operations between them, the output of the last one
// at its true scale.  Returns the carried exponents k_l (all zero = nothing changed); *min_exp = the smallest floor(log2 max|W_l'|) over
// the segment's non-zero layers after rebalancing (callers compare it with REB_LO - something to refuse a blob).
inline std::vector<int> rebalance_chain(const std::vector<RebLayer> &layers, int *min_exp = nullptr) {
    const int n = (int)layers.size();
    std::vector<int> ks((size_t)n, 0);
    int kprev = 0;
    for (int l = 0; l < n; ++l) {
        const int e = reb_exponent(*layers[l].w);
        int k = 0;
        if (l + 1 < n && e > -100000) {
            const int u = e - kprev;
            if (u < REB_LO || u > REB_HI) k = REB_MID - u;
        }
        const int dw = k - kprev;
        if (dw != 0) for (float &v : *layers[l].w) v = ldexpf(v, dw);
        if (k != 0 && layers[l].b) for (float &v : *layers[l].b) v = ldexpf(v, k);
        if (min_exp && e > -100000 && e + dw < *min_exp) *min_exp = e + dw;
        ks[(size_t)l] = k;
        kprev = k;
    }
    return ks;
}

}  // namespace vadx
