// valu_rate.hip -- what a wave64 VALU instruction costs on gfx950, measured (developer tool; `hipcc --offload-arch=gfx950
// -O2 tools/valu_rate.hip -o tools/_prof/valu_rate`, run on the GPU box).
//
// Each wave runs N iterations of a block of 8 x 8 = 64 instructions of ONE kind on 8 independent accumulators (no
// dependent-issue stalls inside a block) and reports core cycles (s_memtime) per instruction; launched with W waves per SIMD
// on every SIMD of the chip (256 CUs x 4) or on one SIMD only.  Both clocks are read: core cycles / wall time = the
// frequency the chip sustains under that load.  DEP variants: one accumulator, i.e. the dependent-issue latency.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(1024) k(int n, long long* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
  double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
  const float b = 1.0000001f, c = 1e-9f;
  const f2 pb = {b, b}, pc = {c, c};
  const double db = 1.0000001, dc = 1e-9;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    if (KIND == 0) {  // v_fma_f32, 8 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
    } else if (KIND == 2) {  // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
    } else if (KIND == 3) {  // v_fma_f64
      REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));)
    } else if (KIND == 4) {  // v_fma_f32, ONE dependent chain
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                        : "+v"(a0) : "v"(b), "v"(c));)
    } else if (KIND == 5) {  // v_pk_fma_f32, ONE dependent chain
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                        "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2"
                        : "+v"(p0) : "v"(pb), "v"(pc));)
    } else if (KIND == 6) {  // v_fma_f64, ONE dependent chain
      REP8(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                        "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2"
                        : "+v"(d0) : "v"(db), "v"(dc));)
    } else if (KIND == 7) {  // v_rcp_f32 (transcendental rate), 8 chains
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 8) {  // v_mov_b32 with DPP (quad_perm), 8 chains
      REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
            (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if ((threadIdx.x & 63) == 0) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = (w1 - w0) + (s == 12345.f ? 1 : 0);
  }
}

template <int KIND>
void run(const char* name, int waves_per_simd, bool whole_chip, long long* d_out) {
  const int n = 2000;
  // one block = 4 x waves_per_simd waves = waves_per_simd on every SIMD of ONE CU; whole chip: one such block per CU
  const int blocks = whole_chip ? 256 : 1, nw = blocks * 4 * waves_per_simd;
  (void)hipMemset(d_out, 0, sizeof(long long) * 2 * nw);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256 * waves_per_simd), 0, 0, 64, d_out, 1.f);  // warm-up
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256 * waves_per_simd), 0, 0, n, d_out, 1.f);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(2 * nw);
  (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int w = 0; w < nw; ++w) { cyc += (double)h[2 * w]; wall += (double)h[2 * w + 1]; }
  cyc /= nw; wall /= nw;
  const double instr = (double)n * 64;
  printf("%-28s %d wave(s)/SIMD %-10s  %6.2f core cycles per instruction per wave  -> %5.2f cycles of SIMD issue per instruction;  %6.0f MHz (s_memtime per s_memrealtime)\n",
         name, waves_per_simd, whole_chip ? "whole chip" : "one CU", cyc / instr, cyc / instr / waves_per_simd, cyc / (wall / 100.0));
}

int main() {
  long long* d_out;
  (void)hipMalloc(&d_out, sizeof(long long) * 2 * 256 * 8 * 4);
  for (int chip = 0; chip < 2; ++chip)
    for (int w : {1, 2, 4}) {
      run<0>("v_fma_f32 x8 independent", w, chip, d_out);
      run<1>("v_pk_fma_f32 x8 independent", w, chip, d_out);
      run<2>("v_pk_mul_f32 x8 independent", w, chip, d_out);
      run<3>("v_fma_f64 x8 independent", w, chip, d_out);
      run<7>("v_rcp_f32 x8 independent", w, chip, d_out);
      run<8>("v_mov_b32_dpp x8 independent", w, chip, d_out);
      if (w == 1) {
        run<4>("v_fma_f32 dependent chain", w, chip, d_out);
        run<5>("v_pk_fma_f32 dependent chain", w, chip, d_out);
        run<6>("v_fma_f64 dependent chain", w, chip, d_out);
      }
    }
  return 0;
}
