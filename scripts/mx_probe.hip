// mx_probe -- operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, E8M0 block scales), checked on
// the MI355X against a host evaluation: the layout is documented in cdna4_isa.md, which is not on this box, so the MX GEMM kernel
// (csrc/gemm.hip: gemm_mx8_tile_kernel) rests on what this program verifies.
//   hipcc --offload-arch=gfx950 -O2 scripts/mx_probe.hip -o whisperjav_amd/csrc/mx_probe
// Hypotheses per operand: HYP 0: lane l holds row l & 15, bytes k = 32 (l >> 4) .. + 31 (one MX block per lane);
//                         HYP 1: two halves: k = 16 (l >> 4) .. + 15 and 64 + 16 (l >> 4) .. + 15.
// The lane's scale (E8M0 byte 0 of the scale VGPR, op_sel 0) applies to the lane's 32 elements.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void one_mfma(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D, int hyp) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 av, bv;
  for (int w = 0; w < 8; ++w) {
    int k = hyp == 0 ? g * 32 + w * 4 : (w < 4 ? g * 16 + w * 4 : 64 + g * 16 + (w - 4) * 4);      // hyp 1 and 2 load alike
    av[w] = *reinterpret_cast<const int*>(A + r * 128 + k);
    bv[w] = *reinterpret_cast<const int*>(B + r * 128 + k);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const int sa = SA[r * 4 + g], sb = SB[r * 4 + g];
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
  for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];      // lane l, register i
}

static float e4m3(uint8_t v) {      // OCP e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

// k -> lane group that holds element k under an operand hypothesis
static int group_of(int hyp, int k) { return hyp == 0 ? k / 32 : hyp == 1 ? (k % 64) / 16 : k / 32; }      // hyp 2: data as 1, scale of MX block k / 32 from lane group k / 32

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128), SA(64), SB(64);
  srand(7);
  // small integers (exact in e4m3: 0x38 = 1.0, 0x40 = 2.0, 0x44 = 3.0, 0x48 = 4.0 and negatives), scales 2^-1 .. 2^2: every product and
  // every partial sum is an exactly representable float, so the right layout gives ZERO error
  const uint8_t vals[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0xb8, 0xc0, 0xc4, 0xc8};
  for (auto& v : A) v = vals[rand() % 9];
  for (auto& v : B) v = vals[rand() % 9];
  const bool unit = getenv("MX_UNIT_SCALES") != nullptr;
  for (auto& v : SA) v = unit ? 127 : 126 + rand() % 4;      // E8M0: 2^(v - 127)
  for (auto& v : SB) v = unit ? 127 : 126 + rand() % 4;
  uint8_t *dA, *dB, *dSA, *dSB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 3; ++hyp) {
    // reference under this hypothesis: the scale of element k of row i is the scale byte of the lane group that holds k
    double M[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 128; ++k) {
        const int g = group_of(hyp, k);
        s += (double)e4m3(A[i * 128 + k]) * std::ldexp(1.0, SA[i * 4 + g] - 127) * e4m3(B[j * 128 + k]) * std::ldexp(1.0, SB[j * 4 + g] - 127);
      }
      M[i][j] = s;
    }
    one_mfma<<<1, 64>>>(dA, dB, dSA, dSB, dD, hyp);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      const double v = D[l * 4 + i];
      bad_a += v != M[4 * (l >> 4) + i][l & 15];      // D[lane][i] = M[a-row 4 (l >> 4) + i][b-row l & 15]
      bad_b += v != M[l & 15][4 * (l >> 4) + i];      // transposed
    }
    if (getenv("MX_VERBOSE")) {
      for (int l = 0; l < 64; l += 21) printf("lane %d: D = %g %g %g %g | M[4g+i][r] = %g %g %g %g | M[r][4g+i] = %g %g %g %g\n", l, D[l * 4], D[l * 4 + 1], D[l * 4 + 2], D[l * 4 + 3],
             M[4 * (l >> 4)][l & 15], M[4 * (l >> 4) + 1][l & 15], M[4 * (l >> 4) + 2][l & 15], M[4 * (l >> 4) + 3][l & 15],
             M[l & 15][4 * (l >> 4)], M[l & 15][4 * (l >> 4) + 1], M[l & 15][4 * (l >> 4) + 2], M[l & 15][4 * (l >> 4) + 3]);
    }
    printf("{\"mx_probe\": \"lane (r, g) holds %s; its scale byte applies to those 32 elements\", \"mismatches_out_a_of_256\": %d, "
           "\"mismatches_out_b_of_256\": %d}\n", hyp == 0 ? "k = 32 g .. 32 g + 31" : hyp == 1 ? "k = 16 g .. + 15 and 64 + 16 g .. + 15" : "k = 16 g .. + 15 and 64 + 16 g .. + 15, but the scale byte of lane group g belongs to MX block g = k / 32 (THE LAYOUT)", bad_a, bad_b);
  }
  return 0;
}
