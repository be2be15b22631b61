// valu_rate.hip -- what the integer VALU really issues on MI355X: wave-instructions per cycle per SIMD for the instruction
// kinds of the overlap-DP loop (v_add_u32, v_max3_i32, v_and_or_b32, v_cmp + v_cndmask, DPP move), at 1, 2, 4 and 8 waves
// per SIMD, and the shader clock under that load (s_memtime ticks per s_memrealtime tick x 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_rate_probe tools/probes/valu_rate.hip && tools/probes/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k_rate(int iters, int *out, unsigned long long *clk)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    int b = threadIdx.x ^ 0x55, c = blockIdx.x + 3;
    int sc; asm volatile("s_mov_b32 %0, 0xfff80000" : "=s"(sc));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] + b;                                                   // v_add_u32
                else if (OP == 1) a[i] = max(max(a[i], b), c);                                  // v_max3_i32
                else if (OP == 2) a[i] = (a[i] & 0xFFFCFFFF) | b;                               // v_and_or_b32
                else if (OP == 3) a[i] = (a[i] == c) ? b : (a[i] + 1);                          // v_cmp + v_cndmask (+ add)
                else if (OP == 4) a[i] = __builtin_amdgcn_update_dpp(a[i], a[i], 0x138, 0xf, 0xf, false) + 1;   // DPP wave_shr + add
                else if (OP == 5) { a[i] = a[i] + b; a[i] = max(max(a[i], b), c) & 0xFFFCFFFF; }   // the chain: add, max3, and (dependent)
                else if (OP == 6) asm volatile("v_add_u32 %0, 0xfff80000, %0" : "+v"(a[i]));     // 32-bit literal: 8-byte encoding
                else if (OP == 7) asm volatile("v_and_b32 %0, 0xfffcffff, %0" : "+v"(a[i]));
                else if (OP == 8) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));    // the constant in an SGPR
                else if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                else if (OP == 10) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                else if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                else if (OP == 12) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 13) asm volatile("v_add_u32 %0, 64, %0" : "+v"(a[i]));              // inline constant
                else if (OP == 14) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");   // back to back: the hardware interlocks (or the assembler objects)
                asm volatile("" : "+v"(a[i]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP> static void run(const char *name, int per_inst)
{
    int *out; unsigned long long *clk;
    const int iters = 4000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;                     // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_rate<OP><<<blocks, 256>>>(10, out, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_rate<OP><<<blocks, 256>>>(iters, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double mhz = 0; for (int b = 0; b < blocks; b++) mhz += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0; mhz /= blocks;
        const double insts = (double)blocks * 4 * iters * 64.0 * per_inst;                      // wave-instructions
        const double per_simd_per_s = insts / (ms * 1e-3) / 1024.0;
        printf("%-28s waves/SIMD %d: %8.3f ms  %.3f G wave-inst/s/SIMD  = one per %.2f cycles at the measured %.0f MHz (s_memtime)\n",
               name, wps, ms, per_simd_per_s / 1e9, mhz * 1e6 / per_simd_per_s, mhz);
        hipFree(out); hipFree(clk);
    }
}

int main()
{
    run<0>("v_add_u32", 1);
    run<1>("v_max3_i32", 1);
    run<2>("v_and_or_b32", 1);
    run<3>("v_cmp+v_cndmask+v_add", 3);
    run<4>("v_mov_dpp wave_shr + v_add", 2);
    run<5>("add, max3, and (dependent)", 3);
    run<6>("v_add_u32 literal", 1);
    run<7>("v_and_b32 literal", 1);
    run<8>("v_add_u32 sgpr", 1);
    run<9>("v_mov_b32", 1);
    run<10>("v_cmp_eq_u32 vcc", 1);
    run<11>("v_cndmask_b32 vcc", 1);
    run<12>("v_max_i32", 1);
    run<13>("v_add_u32 inline const", 1);
    run<14>("v_cmp + v_cndmask adjacent", 2);
    return 0;
}
