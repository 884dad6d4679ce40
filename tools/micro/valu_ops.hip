// Issue rate per VALU opcode class on gfx950: cycles per wave64 instruction per SIMD with
// 1 / 2 / 4 / 8 waves per SIMD resident (8 independent dependency chains per wave).
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP3(NAME, INSN)                                                                        \
  __global__ __launch_bounds__(1024) void NAME(unsigned* out, int iters) {                     \
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 9, f = c + 11,   \
             g = d + 13, h = e + 15;                                                           \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                         \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(e));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(d) : "v"(e), "v"(f));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(e) : "v"(f), "v"(g));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(g) : "v"(h), "v"(a));                       \
        asm volatile(INSN " %0, %0, %1, %2" : "+v"(h) : "v"(a), "v"(b));                       \
      }                                                                                        \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;                \
  }

#define OP2(NAME, INSN)                                                                        \
  __global__ __launch_bounds__(1024) void NAME(unsigned* out, int iters) {                     \
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 9, f = c + 11,   \
             g = d + 13, h = e + 15;                                                           \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                         \
        asm volatile(INSN " %0, %0, %1" : "+v"(a) : "v"(b));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(b) : "v"(c));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(c) : "v"(d));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(d) : "v"(e));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(e) : "v"(f));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(f) : "v"(g));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(g) : "v"(h));                                   \
        asm volatile(INSN " %0, %0, %1" : "+v"(h) : "v"(a));                                   \
      }                                                                                        \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;                \
  }

// DPP move: v_mov_b32 with a wave shift (what kpe_main uses for its neighbours)
#define OPD(NAME, INSN, MOD)                                                                   \
  __global__ __launch_bounds__(1024) void NAME(unsigned* out, int iters) {                     \
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = b + 9, f = c + 11,   \
             g = d + 13, h = e + 15;                                                           \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                         \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(a) : "v"(b));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(b) : "v"(c));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(c) : "v"(d));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(d) : "v"(e));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(e) : "v"(f));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(f) : "v"(g));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(g) : "v"(h));                              \
        asm volatile(INSN " %0, %0, %1 " MOD : "+v"(h) : "v"(a));                              \
      }                                                                                        \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;                \
  }

OP3(k_max3_u32, "v_max3_u32")
OP3(k_med3_u32, "v_med3_u32")
OP3(k_max3_f32, "v_max3_f32")
OP3(k_fma_f32, "v_fma_f32")
OP3(k_perm_b32, "v_perm_b32")
OP3(k_bfi_b32, "v_bfi_b32")
OP3(k_and_or_b32, "v_and_or_b32")
OP3(k_lshl_or_b32, "v_lshl_or_b32")
OP3(k_add3_u32, "v_add3_u32")
OP3(k_sad_u8, "v_sad_u8")
OP3(k_alignbit, "v_alignbit_b32")
OP3(k_pk_max3_f16, "v_pk_maximum3_f16")
OP3(k_pk_min3_f16, "v_pk_minimum3_f16")
OP3(k_max3_i16, "v_max3_i16")
OP3(k_med3_i16, "v_med3_i16")
OP2(k_max_u32, "v_max_u32")
OP2(k_min_u32, "v_min_u32")
OP2(k_and_b32, "v_and_b32")
OP2(k_xor_b32, "v_xor_b32")
OP2(k_add_u32, "v_add_u32")
OP2(k_lshlrev, "v_lshlrev_b32")
OP2(k_pk_max_u16, "v_pk_max_u16")
OP2(k_pk_min_u16, "v_pk_min_u16")
OP2(k_pk_add_u16, "v_pk_add_u16")
OP2(k_pk_max_i16, "v_pk_max_i16")
OP2(k_pk_max_f16, "v_pk_max_f16")
OP2(k_max_u16, "v_max_u16")
OP2(k_max_f32, "v_max_f32")
OP2(k_add_f32, "v_add_f32")
OP2(k_pk_add_f32x, "v_max_f16")
OPD(k_max_u32_dpp_shr, "v_max_u32_dpp", "wave_shr:1")
OPD(k_max_u32_dpp_row, "v_max_u32_dpp", "row_shr:1")
OPD(k_max_u32_dpp_quad, "v_max_u32_dpp", "quad_perm:[1,0,3,2]")
OPD(k_max_u32_sdwa, "v_max_u32_sdwa", "dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2")
OPD(k_max_u16_sdwa, "v_max_u16_sdwa", "dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")

// 64-bit forms: %0 is a register pair
#define OP64(NAME, TEXT)                                                                       \
  __global__ void NAME(unsigned* out, int iters) {                                             \
    unsigned long long a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;           \
    unsigned x = threadIdx.x & 31, y = x + 1;                                                  \
    for (int it = 0; it < iters; ++it) {                                                       \
      _Pragma("unroll") for (int u = 0; u < 32; ++u) {                                         \
        asm volatile(TEXT : "+v"(a) : "v"(x), "v"(y) : "vcc");                                 \
        asm volatile(TEXT : "+v"(b) : "v"(y), "v"(x) : "vcc");                                 \
        asm volatile(TEXT : "+v"(c) : "v"(x), "v"(y) : "vcc");                                 \
        asm volatile(TEXT : "+v"(d) : "v"(y), "v"(x) : "vcc");                                 \
      }                                                                                        \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a ^ b ^ c ^ d);                    \
  }
OP64(k_lshlrev_b64, "v_lshlrev_b64 %0, %1, %0")
OP64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
OP64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
OP2(k_mul_lo_u32, "v_mul_lo_u32")
OP2(k_mul_hi_u32, "v_mul_hi_u32")
OP2(k_mul_u32_u24, "v_mul_u32_u24")
OP3(k_mad_u32_u24, "v_mad_u32_u24")
OP3(k_bitop3, "v_xad_u32")

typedef void (*kern_t)(unsigned*, int);
struct Entry { const char* name; kern_t k; };

int main() {
  unsigned* d; (void)hipMalloc(&d, 256 * 2048 * 4 * 2);
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  printf("clock %d kHz, CUs %d\n", pr.clockRate, pr.multiProcessorCount);
  Entry es[] = {
#define E(x) {#x, x}
      E(k_max3_u32), E(k_med3_u32), E(k_max3_f32), E(k_fma_f32), E(k_perm_b32), E(k_bfi_b32),
      E(k_and_or_b32), E(k_lshl_or_b32), E(k_add3_u32), E(k_sad_u8), E(k_alignbit),
      E(k_pk_max3_f16), E(k_pk_min3_f16), E(k_max3_i16), E(k_med3_i16),
      E(k_max_u32), E(k_min_u32), E(k_and_b32), E(k_xor_b32), E(k_add_u32), E(k_lshlrev),
      E(k_pk_max_u16), E(k_pk_min_u16), E(k_pk_add_u16), E(k_pk_max_i16), E(k_pk_max_f16),
      E(k_max_u16), E(k_max_f32), E(k_add_f32), E(k_pk_add_f32x),
      E(k_max_u32_dpp_shr), E(k_max_u32_dpp_row), E(k_max_u32_dpp_quad), E(k_max_u32_sdwa),
      E(k_max_u16_sdwa), E(k_lshlrev_b64), E(k_mad_u64_u32), E(k_lshl_add_u64), E(k_mul_lo_u32), E(k_mul_hi_u32),
      E(k_mul_u32_u24), E(k_mad_u32_u24), E(k_bitop3),
  };
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (auto& e : es) {
    printf("%-22s", e.name);
    for (int threads : {256, 512, 1024, 2048}) {  // per CU: 1, 2, 4, 8 waves per SIMD
      int bs = threads > 1024 ? 1024 : threads, blocks = 256 * (threads / bs);
      int iters = 1000;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(bs), 0, 0, d, 10);
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(bs), 0, 0, d, iters);
      (void)hipEventRecord(b); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      double winstr = (double)blocks * bs / 64 * iters * 128;
      double per_cyc = winstr / (ms * 1e6) / 1024 / 2.4;
      printf("  w%d: %.2f cyc", threads / 256, 1.0 / per_cyc);
    }
    printf("   (cycles per wave64 instruction per SIMD at a nominal 2.4 GHz)\n");
    fflush(stdout);
  }
  return 0;
}
