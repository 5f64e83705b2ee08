// What bounds the Fr-heavy kernels (zk_ntt_*, zk_rslb_chunks, the MSMs): the multiplier.  
//   part A  sustained issue rate on gfx950 of the instructions a Montgomery product can be built from
//           (v_mad_u64_u32, v_mul_lo/hi_u32, v_mul_u32_u24, v_fma_f64, v_add_f64, 64-bit integer add, v_fma_f32 as the yardstick)
//   part B  Montgomery products per second: csrc/zkwg_fr.h's fr_mont_mul (8 x 32-bit CIOS, 128 v_mad_u64_u32)
//           against a 5 x 52-bit-limb product on the FP64 FMA pipe (Emmart-Weems hi/lo split, round-toward-zero),
//           operands kept in limb form between products (values < 2r, R = 2^260: no conditional subtraction)
//   part C  the FP64 product bit-exact against fr_mont_mul on 2^20 random pairs + the corner values 0, 1, r - 1, R mod r
//   hipcc -O3 --offload-arch=gfx950 -I zk-email-verify_amd/csrc tools/mulbench.hip -o tools/mulbench && tools/mulbench [out.json]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "zkwg_fr.h"
#include "zkwg_fr52.h"
#include "zkwg_fr29.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---------------- part A: raw issue rates -----------------------------------------------------------------------------
// 8 independent dependency chains per lane, 8 instructions per chain per loop iteration: 64 instructions per iteration.
#define REP8(x) x x x x x x x x
enum { OP_MAD64 = 0, OP_MULLO, OP_MULHI, OP_MUL24, OP_MULHI24, OP_FMA64, OP_ADD64F, OP_ADDI64, OP_FMA32, OP_MAD24, OP_LSHR64, OP_LSHLADD64, N_OPS };
static const char* op_name[N_OPS] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_fma_f64",
                                     "v_add_f64", "v_add_co_u32+v_addc_co_u32", "v_fma_f32", "v_mad_u32_u24", "v_lshrrev_b64", "v_lshl_add_u64"};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(u64* out, u32 iters, u32 seed) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  u32 a = seed * 2654435761u + tid, b = (seed ^ 0x9e3779b9u) + tid * 3u;
  u64 c[8];
  for (int k = 0; k < 8; ++k) c[k] = ((u64)(a + k) << 32) | (b + k);
  if (OP == OP_MAD64) {
    for (u32 i = 0; i < iters; ++i) {
      REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                        "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");)
    }
  } else if (OP == OP_LSHR64) {
    for (u32 i = 0; i < iters; ++i) {
      REP8(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n"
                        "v_lshrrev_b64 %4, 1, %4\n v_lshrrev_b64 %5, 1, %5\n v_lshrrev_b64 %6, 1, %6\n v_lshrrev_b64 %7, 1, %7\n"
                        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));)
    }
  } else if (OP == OP_LSHLADD64) {
    const u64 inc = ((u64)a << 32) | b;
    for (u32 i = 0; i < iters; ++i) {
      REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                        "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(inc));)
    }
  } else if (OP == OP_FMA64 || OP == OP_ADD64F) {
    double d[8], x = 1.0 + 1e-9 * (double)(tid & 7), y = 1e-12;
    for (int k = 0; k < 8; ++k) d[k] = 1.0 + k;
    for (u32 i = 0; i < iters; ++i) {
      if (OP == OP_FMA64) {
        REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                          "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                          : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(x), "v"(y));)
      } else {
        REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                          "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                          : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(y));)
      }
    }
    for (int k = 0; k < 8; ++k) c[k] = (u64)__double_as_longlong(d[k]);
  } else if (OP == OP_ADDI64) {
    u32 lo[8], hi[8];
    for (int k = 0; k < 8; ++k) { lo[k] = (u32)c[k]; hi[k] = (u32)(c[k] >> 32); }
    for (u32 i = 0; i < iters; ++i) {
      // one "instruction" here is the pair v_add_co_u32 / v_addc_co_u32 (a 64-bit add)
#define ADD4(o) asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n" \
                             "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" \
                             : "+v"(lo[o]), "+v"(hi[o]), "+v"(lo[o + 1]), "+v"(hi[o + 1]), "+v"(lo[o + 2]), "+v"(hi[o + 2]), "+v"(lo[o + 3]), "+v"(hi[o + 3]) : "v"(a), "v"(b) : "vcc");
      REP8(ADD4(0) ADD4(4))
    }
    for (int k = 0; k < 8; ++k) c[k] = ((u64)hi[k] << 32) | lo[k];
  } else {
    u32 e[8];
    for (int k = 0; k < 8; ++k) e[k] = a + k;
    float f[8], fx = 1.0f + 1e-7f * (tid & 3), fy = 1e-9f;
    for (int k = 0; k < 8; ++k) f[k] = 1.0f + k;
#define INT8(mn) REP8(asm volatile(mn " %0, %0, %8\n " mn " %1, %1, %8\n " mn " %2, %2, %8\n " mn " %3, %3, %8\n " mn " %4, %4, %8\n " mn " %5, %5, %8\n " mn " %6, %6, %8\n " mn " %7, %7, %8\n" \
                                   : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "v"(b));)
    for (u32 i = 0; i < iters; ++i) {
      if (OP == OP_MULLO) { INT8("v_mul_lo_u32") }
      else if (OP == OP_MULHI) { INT8("v_mul_hi_u32") }
      else if (OP == OP_MUL24) { INT8("v_mul_u32_u24") }
      else if (OP == OP_MULHI24) { INT8("v_mul_hi_u32_u24") }
      else if (OP == OP_MAD24) {
        REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                          "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7\n"
                          : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : "v"(b));)
      } else {
        REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                          : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(fx), "v"(fy));)
      }
    }
    for (int k = 0; k < 8; ++k) c[k] = e[k] ^ (u64)__float_as_uint(f[k]);
  }
  u64 s = 0;
  for (int k = 0; k < 8; ++k) s ^= c[k];
  if (s == 0x123456789abcdefull) out[tid] = s;   // keeps the chains alive, (almost) never stores
}

// ---------------- part B: Montgomery products per second --------------------------------------------------------------
// CH independent product chains per lane (x <- x * y), `iters` products each.
template <int CH>
__global__ __launch_bounds__(256) void prod_cios(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  Fr x[CH];
  const Fr y = in[(tid + 1) & 1023];
  for (int k = 0; k < CH; ++k) x[k] = in[(tid + 7 * k) & 1023];
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = fr_mont_mul_cios32(x[k], y);
  Fr s = x[0];
  for (int k = 1; k < CH; ++k) s = fr_add(s, x[k]);
  out[tid] = s;
}
// the product the kernels call since round 5: 9 x 29-bit product scanning behind the 4 x 64-bit interface (csrc/zkwg_comba29.h)
template <int CH>
__global__ __launch_bounds__(256) void prod_dropin(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  Fr x[CH];
  const Fr y = in[(tid + 1) & 1023];
  for (int k = 0; k < CH; ++k) x[k] = in[(tid + 7 * k) & 1023];
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = fr_mont_mul_comba(x[k], y);
  Fr s = x[0];
  for (int k = 1; k < CH; ++k) s = fr_add(s, x[k]);
  out[tid] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void prod_f52(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  const Fr52Ctx cx = fr52_enter();
  Fr52 x[CH];
  const Fr52 y = fr52_from_fr(in[(tid + 1) & 1023], cx);
  for (int k = 0; k < CH; ++k) x[k] = fr52_from_fr(in[(tid + 7 * k) & 1023], cx);
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = fr52_mul(x[k], y, cx);
  Fr s = fr52_to_fr(x[0], cx);
  for (int k = 1; k < CH; ++k) s = fr_add(s, fr52_to_fr(x[k], cx));
  out[tid] = s;
}
// 32-bit limbs, operand scanning with the carries of a row resolved once per row (8 mads, then one add-with-carry chain)
__device__ __forceinline__ Fr fr_mont_mul_rw(const Fr& a, const Fr& b) {
  const u32 P32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  u32 A[8], Bv[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { A[2 * i] = (u32)a.l[i]; A[2 * i + 1] = (u32)(a.l[i] >> 32); Bv[2 * i] = (u32)b.l[i]; Bv[2 * i + 1] = (u32)(b.l[i] >> 32); }
  u64 t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // t[j] < 2^32 between rows (the high word of the pair stays zero), t[8] the running top
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = (u64)A[j] * Bv[i] + t[j];         // < 2^64
    const u32 m = (u32)d[0] * 0xefffffffu;
    // t' = (sum_j d_j 2^(32 j) + t8 2^256 + m P) / 2^32, resolved with one carry chain
    u64 e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (u64)m * P32[j] + (u32)d[j];      // low halves joined with the reduction row: < 2^64
    u64 c = e[0] >> 32;                                                   // low word of e[0] is zero
#pragma unroll
    for (int j = 1; j < 8; ++j) { c += (u64)(u32)e[j] + (d[j - 1] >> 32); t[j - 1] = (u32)c; c = (c >> 32) + (e[j] >> 32); }
    c += (d[7] >> 32) + t[8];
    t[7] = (u32)c; t[8] = c >> 32;
  }
  Fr r{{t[0] | (t[1] << 32), t[2] | (t[3] << 32), t[4] | (t[5] << 32), t[6] | (t[7] << 32)}};
  if (t[8] || fr_geq(r, fr_p())) { u64 bw; r = fr_sub_raw(r, fr_p(), bw); }
  return r;
}
template <int CH>
__global__ __launch_bounds__(256) void prod_rw(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  Fr x[CH];
  const Fr y = in[(tid + 1) & 1023];
  for (int k = 0; k < CH; ++k) x[k] = in[(tid + 7 * k) & 1023];
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = fr_mont_mul_rw(x[k], y);
  Fr s = x[0];
  for (int k = 1; k < CH; ++k) s = fr_add(s, x[k]);
  out[tid] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void prod_f29(const Fr* in, Fr* out, u32 iters) {
  const u32 tid = blockIdx.x * 256 + threadIdx.x;
  Fr29 x[CH];
  const Fr29 y = fr29_from_fr(in[(tid + 1) & 1023]);
  for (int k = 0; k < CH; ++k) x[k] = fr29_from_fr(in[(tid + 7 * k) & 1023]);
  for (u32 i = 0; i < iters; ++i)
    for (int k = 0; k < CH; ++k) x[k] = fr29_mul(x[k], y);
  Fr s = fr29_to_fr(x[0]);
  for (int k = 1; k < CH; ++k) s = fr_add(s, fr29_to_fr(x[k]));
  out[tid] = s;
}
// out[i] += 1 iff rw(a, b) != cios(a, b); += 2 iff 2^5 * f29(a, b) != cios(a, b) (R = 2^261 against 2^256)
__global__ __launch_bounds__(256) void check_new(const Fr* a, const Fr* b, u32 n, u32* bad) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Fr c = fr_mont_mul_cios32(a[i], b[i]);
  if (!fr_eq(fr_mont_mul_comba(a[i], b[i]), c)) atomicAdd(bad + 3, 1u);
  if (!fr_eq(fr_mont_mul_rw(a[i], b[i]), c)) atomicAdd(bad, 1u);
  Fr f = fr29_to_fr(fr29_mul(fr29_from_fr(a[i]), fr29_from_fr(b[i])));
  for (int k = 0; k < 5; ++k) f = fr_add(f, f);
  if (!fr_eq(f, c)) atomicAdd(bad + 1, 1u);
  // unnormalised first operand: (a + a + a) * b with lazy limb sums
  const Fr29 a29 = fr29_from_fr(a[i]);
  Fr g = fr29_to_fr(fr29_mul(fr29_add(fr29_add(a29, a29), a29), fr29_from_fr(b[i])));
  for (int k = 0; k < 5; ++k) g = fr_add(g, g);
  if (!fr_eq(g, fr_add(fr_add(c, c), c))) atomicAdd(bad + 2, 1u);
}
// ---------------- part C: bit-exactness --------------------------------------------------------------------------------
// out[i] = 1 iff 2^4 * f52(a, b) != cios(a, b) (R = 2^260 against R = 2^256), after full reduction of both
__global__ __launch_bounds__(256) void check_f52(const Fr* a, const Fr* b, u32 n, u32* bad, Fr* first_bad) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Fr c = fr_mont_mul_cios32(a[i], b[i]);
  const Fr52Ctx cx = fr52_enter();
  Fr f = fr52_to_fr(fr52_mul(fr52_from_fr(a[i], cx), fr52_from_fr(b[i], cx), cx), cx);
  for (int k = 0; k < 4; ++k) f = fr_add(f, f);
  if (!fr_eq(f, c)) { if (atomicAdd(bad, 1u) == 0) { first_bad[0] = a[i]; first_bad[1] = b[i]; first_bad[2] = c; first_bad[3] = f; } }
}

template <class F> static float time_ms(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return best;
}
static u64 sm64(u64& s) { u64 z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static Fr rand_fr(u64& s) {
  Fr x{{sm64(s), sm64(s), sm64(s), sm64(s) & 0x3fffffffffffffffull}};
  while (fr_geq(x, fr_p())) { u64 bw; x = fr_sub_raw(x, fr_p(), bw); }
  return x;
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const double clk_ghz = pr.clockRate / 1e6;
  printf("device %s, %d CUs, %.2f GHz max clock\n", pr.name, cus, clk_ghz);
  u64* sink; CK(hipMalloc((void**)&sink, 256ull * 256 * 64 * 8));
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"max_clock_GHz\": %.3f,\n \"issue_rates\": {", pr.name, cus, clk_ghz);
  // part A
  const u32 iters = 2048;
  double rate[N_OPS];
  for (int wps : {1, 2, 4}) {
    const u32 blocks = cus * wps;   // 256 threads = 4 wavefronts = one per SIMD of a CU: wps wavefronts per SIMD
    printf("-- %d wavefront(s) per SIMD: lane-operations per cycle per SIMD at max clock (G lane-ops/s)\n", wps);
    for (int op = 0; op < N_OPS; ++op) {
      float ms = 0;
#define RUN(OPX) case OPX: ms = time_ms([&] { hipLaunchKernelGGL((rate_kernel<OPX>), dim3(blocks), dim3(256), 0, 0, sink, iters, 12345u); }); break;
      switch (op) { RUN(OP_MAD64) RUN(OP_MULLO) RUN(OP_MULHI) RUN(OP_MUL24) RUN(OP_MULHI24) RUN(OP_FMA64) RUN(OP_ADD64F) RUN(OP_ADDI64) RUN(OP_FMA32) RUN(OP_MAD24) RUN(OP_LSHR64) RUN(OP_LSHLADD64) }
      const double ops = (double)blocks * 256.0 * iters * 64.0;
      const double g = ops / ms / 1e6;
      rate[op] = g;
      printf("   %-30s %9.1f G/s  = %5.2f lanes/cycle/SIMD\n", op_name[op], g, g / (cus * 4.0 * clk_ghz));
      if (js && wps == 4) fprintf(js, "%s\"%s\": {\"G_lane_ops_per_s\": %.1f, \"lanes_per_cycle_per_simd\": %.3f}", op ? ", " : "", op_name[op], g, g / (cus * 4.0 * clk_ghz));
    }
  }
  if (js) fprintf(js, "},\n");
  // inputs
  const u32 NCHK = 1u << 20;
  std::vector<Fr> ha(NCHK), hb(NCHK);
  u64 seed = 0x5A4B454D41494Cull;
  for (u32 i = 0; i < NCHK; ++i) { ha[i] = rand_fr(seed); hb[i] = rand_fr(seed); }
  { // corner values in every combination
    Fr rm1 = fr_p(); rm1.l[0] -= 1;
    const Fr cv[6] = {fr_zero(), fr_from_u64(1), rm1, fr_R(), fr_R2(), Fr{{~0ull >> 12, 0, 0, 0}}};
    u32 k = 0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { ha[k] = cv[i]; hb[k] = cv[j]; ++k; }
    for (int i = 0; i < 6; ++i) { ha[k] = cv[i]; hb[k] = rand_fr(seed); ++k; }
  }
  Fr *da, *db, *dout, *dfirst; u32* dbad;
  CK(hipMalloc((void**)&da, NCHK * sizeof(Fr))); CK(hipMalloc((void**)&db, NCHK * sizeof(Fr)));
  CK(hipMalloc((void**)&dout, (size_t)cus * 8 * 256 * sizeof(Fr))); CK(hipMalloc((void**)&dfirst, 4 * sizeof(Fr))); CK(hipMalloc((void**)&dbad, 4));
  CK(hipMemcpy(da, ha.data(), NCHK * sizeof(Fr), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), NCHK * sizeof(Fr), hipMemcpyHostToDevice));
  // part C
  CK(hipMemset(dbad, 0, 4));
  hipLaunchKernelGGL(check_f52, dim3(NCHK / 256), dim3(256), 0, 0, da, db, NCHK, dbad, dfirst);
  CK(hipDeviceSynchronize());
  u32 bad; CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
  printf("-- FP64 52-bit-limb product against fr_mont_mul on %u pairs (42 corner pairs first): %u mismatches\n", NCHK, bad);
  if (bad) {
    Fr fb[4]; CK(hipMemcpy(fb, dfirst, sizeof(fb), hipMemcpyDeviceToHost));
    const char* nm[4] = {"a", "b", "cios", "16*f52"};
    for (int k = 0; k < 4; ++k) printf("   %-7s %016llx %016llx %016llx %016llx\n", nm[k], (unsigned long long)fb[k].l[3], (unsigned long long)fb[k].l[2], (unsigned long long)fb[k].l[1], (unsigned long long)fb[k].l[0]);
  }
  {
    u32* dbad3; CK(hipMalloc((void**)&dbad3, 16)); CK(hipMemset(dbad3, 0, 16));   // [rw, f29, f29 lazy, drop-in]
    hipLaunchKernelGGL(check_new, dim3(NCHK / 256), dim3(256), 0, 0, da, db, NCHK, dbad3);
    CK(hipDeviceSynchronize());
    u32 b3[4]; CK(hipMemcpy(b3, dbad3, 16, hipMemcpyDeviceToHost));
    printf("-- against fr_mont_mul on %u pairs: row-wise 32-bit %u mismatches, 9 x 29-bit Comba %u, with a lazy 3a operand %u, the drop-in (zkwg_comba29.h) %u\n", NCHK, b3[0], b3[1], b3[2], b3[3]);
    if (js) fprintf(js, " \"rw_mismatches\": %u, \"f29_mismatches\": %u, \"f29_lazy_mismatches\": %u,\n", b3[0], b3[1], b3[2]);
    bad += b3[0] + b3[1] + b3[2] + b3[3];
  }
  // part B
  const u32 piters = 512;
  double best_cios = 0, best_f52 = 0, best_rw = 0, best_f29 = 0, best_drop = 0;
  printf("-- Montgomery products per second (chains per lane x wavefronts per SIMD)\n");
  for (int wps : {1, 2, 4, 8}) {
    const u32 blocks = cus * wps;
    float m1 = time_ms([&] { hipLaunchKernelGGL((prod_cios<1>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float m2 = time_ms([&] { hipLaunchKernelGGL((prod_cios<2>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float f1 = time_ms([&] { hipLaunchKernelGGL((prod_f52<1>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float f2 = time_ms([&] { hipLaunchKernelGGL((prod_f52<2>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    const double n1 = (double)blocks * 256 * piters;
    const double c1 = n1 / m1 / 1e6, c2 = 2 * n1 / m2 / 1e6, g1 = n1 / f1 / 1e6, g2 = 2 * n1 / f2 / 1e6;
    float d1 = time_ms([&] { hipLaunchKernelGGL((prod_dropin<1>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float d2 = time_ms([&] { hipLaunchKernelGGL((prod_dropin<2>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    const double dr1 = n1 / d1 / 1e6, dr2 = 2 * n1 / d2 / 1e6;
    if (dr1 > best_drop) best_drop = dr1; if (dr2 > best_drop) best_drop = dr2;
    float r1 = time_ms([&] { hipLaunchKernelGGL((prod_rw<1>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float r2 = time_ms([&] { hipLaunchKernelGGL((prod_rw<2>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float h1 = time_ms([&] { hipLaunchKernelGGL((prod_f29<1>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    float h2 = time_ms([&] { hipLaunchKernelGGL((prod_f29<2>), dim3(blocks), dim3(256), 0, 0, da, dout, piters); });
    const double rw1 = n1 / r1 / 1e6, rw2 = 2 * n1 / r2 / 1e6, f291 = n1 / h1 / 1e6, f292 = 2 * n1 / h2 / 1e6;
    if (rw1 > best_rw) best_rw = rw1; if (rw2 > best_rw) best_rw = rw2;
    if (f291 > best_f29) best_f29 = f291; if (f292 > best_f29) best_f29 = f292;
    printf("   %d waves/SIMD: cios 1 chain %7.1f G/s, 2 chains %7.1f | f52 %7.1f, %7.1f | row-wise %7.1f, %7.1f | 9x29 %7.1f, %7.1f | drop-in %7.1f, %7.1f\n", wps, c1, c2, g1, g2, rw1, rw2, f291, f292, dr1, dr2);
    if (c1 > best_cios) best_cios = c1; if (c2 > best_cios) best_cios = c2;
    if (g1 > best_f52) best_f52 = g1; if (g2 > best_f52) best_f52 = g2;
  }
  const double mad_bound = rate[OP_MAD64] / 128.0;
  printf("-- bound from the measured v_mad_u64_u32 rate: %.1f G products/s (128 per product); cios reaches %.1f (%.2f), f52 %.1f (%.2f x cios), row-wise %.1f (%.2f x), 9x29 %.1f (%.2f x), drop-in %.1f (%.2f x)\n",
         mad_bound, best_cios, best_cios / mad_bound, best_f52, best_f52 / best_cios, best_rw, best_rw / best_cios, best_f29, best_f29 / best_cios, best_drop, best_drop / best_cios);
  if (js) {
    fprintf(js, " \"f52_mismatches\": %u, \"f52_pairs_checked\": %u,\n \"products_G_per_s\": {\"cios_8x32\": %.2f, \"f52_5x52\": %.2f, \"rowwise_8x32\": %.2f, \"comba_9x29\": %.2f, \"comba_9x29_behind_4x64_interface\": %.2f, \"mad_u64_u32_bound_128\": %.2f}\n}\n", bad, NCHK, best_cios, best_f52, best_rw, best_f29, best_drop, mad_bound);
    fclose(js);
  }
  return bad ? 1 : 0;
}
