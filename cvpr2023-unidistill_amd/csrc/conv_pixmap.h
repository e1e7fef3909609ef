// Pixel maps shared by the bf16 and fp32 1x1 kernels (conv2d.hip, conv2d_f32.hip): virtual [P'][K'] matrix -> physical
// channels-last tensor.  Not part of the C ABI (the ABI passes the nine ints of a map).
#ifndef UD_CONV_PIXMAP_H_
#define UD_CONV_PIXMAP_H_
#include <cstddef>

namespace {

// Row p, element k of a VIRTUAL channels-last matrix [P'][K'] -> element offset in the physical tensor.  Lets the 1x1
// kernels (forward / data gradient / weight gradient) run the convolutions whose im2col is a pure permutation:
//   mode 0  identity: p * K + k
//   mode 1  space-to-depth of x[B,H,W,C] with block s (conv k = s, stride s: K' = s*s*C, k = (dy, dx, c); the output
//           side of a transposed conv k = s, stride s is the same map): p = (b, oy, ox) ->
//           ((b*H + s*oy + k / (s*C)) * W + s*ox) * C + k % (s*C)
//   mode 2  spatial subsampling by s (1x1 conv with stride s: K' = C): ((b*H + s*oy) * W + s*ox) * C + k
//   mode 3  im2col of a 3x3 / pad 1 / stride s convolution (input side only; K' = 9*C, k = (tap, c)):
//           pixel (s*oy + ty - 1, s*ox + tx - 1), kNoPixel outside the tensor (the kernels read zeros there)
//   mode 4  data gradient of the same convolution (s = 2) for the input pixels of one parity class (a, b) = (y & 1,
//           x & 1): row p = (batch, i, j) is input pixel (2i + a, 2j + b); an even coordinate is reached by the centre
//           tap only, an odd one by taps 0 and 2, so K' = (1 + a)(1 + b) * C with k = (jy, jx, c) and the source
//           pixel of dy [B,H,W,C] is (i + a*(1 - jy), j + b*(1 - jx)), kNoPixel outside (input side only)
// Modes 2 and 4 address pixel (s*oy + a, s*ox + b): (a, b) is the class offset (0, 0 for a plain strided 1x1).
constexpr size_t kNoPixel = ~(size_t)0;
struct PixMap {
  int mode, s, Ho, Wo, H, W, C, a, b;
  __device__ __forceinline__ size_t off(long long p, int k, int K) const {
    if (mode == 0) return (size_t)p * K + k;
    // rows < 2^30 (every launcher checks): 32-bit divisions (a 64-bit one is a ~100-instruction loop per DMA piece)
    const unsigned pu = (unsigned)p, t = pu / (unsigned)Wo;
    const int ox = (int)(pu - t * (unsigned)Wo);
    const int b = (int)(t / (unsigned)Ho), oy = (int)(t - (unsigned)b * (unsigned)Ho);
    if (mode == 4) {
      const int tapi = k / C, c = k - tapi * C, nx = 1 + this->b;
      const int jy = tapi / nx, jx = tapi - jy * nx;
      const int sy = oy + a * (1 - jy), sx = ox + this->b * (1 - jx);
      if (sy >= H || sx >= W) return kNoPixel;
      return ((size_t)(b * H + sy) * W + sx) * C + c;
    }
    if (mode == 3) {
      const int tap = k / C, c = k - tap * C;
      const int iy = s * oy + tap / 3 - 1, ix = s * ox + tap % 3 - 1;
      if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return kNoPixel;
      return ((size_t)(b * H + iy) * W + ix) * C + c;
    }
    if (mode == 1) {
      const int sc = s * C, dy = k / sc, r = k - dy * sc;
      return ((size_t)(b * H + s * oy + dy) * W + (size_t)s * ox) * C + r;
    }
    return ((size_t)(b * H + s * oy + a) * W + (size_t)s * ox + this->b) * C + k;
  }
};

// nine ints {mode, s, Ho, Wo, H, W, C, a, b} -> PixMap; `slice` = channels per K slice of the kernel that will read through
// the map (64 bf16 / 32 fp32: a slice must stay inside one tap / one class tap)
inline bool map_from_ints(const int* m, PixMap* out, int slice) {
  *out = PixMap{};
  if (!m || m[0] == 0) return true;
  *out = PixMap{m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]};
  if (!((m[0] >= 1 && m[0] <= 4) && m[1] >= 1 && m[2] > 0 && m[3] > 0 && m[6] > 0 && m[6] % 4 == 0)) return false;
  if (m[7] < 0 || m[8] < 0 || m[7] >= m[1] || m[8] >= m[1] || (m[0] != 2 && m[0] != 4 && (m[7] || m[8]))) return false;
  if (m[0] == 3) return m[6] % slice == 0 && m[4] > 0 && m[5] > 0;                  // a slice inside one tap
  if (m[0] == 4) return m[1] == 2 && m[6] % slice == 0 && m[4] > 0 && m[5] > 0;
  if (m[0] == 1) return m[4] >= m[1] * m[2] && m[5] >= m[1] * m[3];            // every s x s block inside the tensor
  return m[4] > m[1] * (m[2] - 1) + m[7] && m[5] > m[1] * (m[3] - 1) + m[8];       // every sampled pixel inside
}

}  // namespace
#endif  // UD_CONV_PIXMAP_H_
