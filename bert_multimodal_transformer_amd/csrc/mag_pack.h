// MAG's master weights (reference layout, /root/reference/modeling.py:13-17: W_hv [H][V+H], W_ha [H][A+H], W_v [H][V], W_a [H][A])
// -> the three operands of the regrouped GEMMs (mag.hip):  We [2H][H] = [W_hv[:, V:] ; W_ha[:, A:]],  Wv [2H][Vp] = [W_hv[:, :V] ; W_v],
// Wa [2H][Ap] = [W_ha[:, :A] ; W_a], modality columns zero-padded to Vp / Ap.  A device function because two kernels run it: the
// stand-alone pack launch (mag.hip) and the step prologue (rowops.hip), whose extra blocks do it while the others wait for PCIe.
#pragma once
#include "kernels.h"

namespace mb {

template <class T>
__device__ __forceinline__ void mag_pack_w_range(const float* __restrict__ W_hv, const float* __restrict__ W_ha,
                                                 const float* __restrict__ W_v, const float* __restrict__ W_a, T* __restrict__ We,
                                                 T* __restrict__ Wv, T* __restrict__ Wa, const MagDims& d, size_t first, size_t stride) {
    const int H = d.H, V = d.V, A = d.A, Vp = d.Vp, Ap = d.Ap;
    const size_t nWe = (size_t)2 * H * H, nWv = (size_t)2 * H * Vp, nWa = (size_t)2 * H * Ap;
    for (size_t i = first; i < nWe + nWv + nWa; i += stride) {
        if (i < nWe) {
            const int j = (int)(i / H), c = (int)(i % H);
            We[i] = from_f<T>(j < H ? W_hv[(size_t)j * (V + H) + V + c] : W_ha[(size_t)(j - H) * (A + H) + A + c]);
        } else if (i < nWe + nWv) {
            const size_t k = i - nWe;
            const int j = (int)(k / Vp), c = (int)(k % Vp);
            float v = 0.f;
            if (c < V) v = j < H ? W_hv[(size_t)j * (V + H) + c] : W_v[(size_t)(j - H) * V + c];
            Wv[k] = from_f<T>(v);
        } else {
            const size_t k = i - nWe - nWv;
            const int j = (int)(k / Ap), c = (int)(k % Ap);
            float v = 0.f;
            if (c < A) v = j < H ? W_ha[(size_t)j * (A + H) + c] : W_a[(size_t)(j - H) * A + c];
            Wa[k] = from_f<T>(v);
        }
    }
}

}  // namespace mb
