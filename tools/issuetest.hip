// issuetest.hip -- cycles per instruction of ONE wave on gfx950: dependent vs independent chains of the VALU/LDS/
// cross-lane instructions the serial entropy kernels are made of.   hipcc --offload-arch=gfx950 -O3 tools/issuetest.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE> __global__ void k(uint32_t *out, int iters, uint64_t *clk, uint32_t seed)
{
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = threadIdx.x;
    lds[threadIdx.x + 64] = threadIdx.x;
    uint4 q4 = {0, 0, 0, 0};
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0;
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 5 + 2, d = seed * 7 + 3, m = 0x9E3779B1u;
    __syncthreads();
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }                       // dependent add
        if (MODE == 1) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }   // 4 independent
        if (MODE == 2) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 3) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 4) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 5) { REP64(asm volatile("v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(m), "v"(b) : "vcc");) }
        if (MODE == 6) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
        if (MODE == 7) { REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %1" : "+v"(a) : "v"(m) : "s20");) }
        if (MODE == 8) { REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a));) a &= 0xfc; }
        if (MODE == 9) { REP64(asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 10) { REP64(asm volatile("v_cmp_ge_u32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 %0, s20, %0" : "+v"(a) : "v"(m) : "vcc", "s20", "scc");) }
        if (MODE == 11) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(b));) }
        if (MODE == 12) { REP64(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(b));) }
        if (MODE == 13) { REP64(asm volatile("s_add_u32 s20, s20, s21\n" ::: "s20", "scc");) }
        if (MODE == 14) { REP64(asm volatile("s_mul_hi_u32 s20, s20, s21\n" ::: "s20");) }
        if (MODE == 15) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 3\n v_mov_b32 %0, s20" : "+v"(a) :: "s20", "scc");) }
        if (MODE == 16) { REP64(asm volatile("ds_write_b32 %1, %0\n v_add_u32 %0, %0, %2" : "+v"(a) : "v"(b & 0xfc), "v"(m));) }
        if (MODE == 18) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 19) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_lshrrev_b32 %0, 1, %0\n v_or_b32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 20) { REP64(asm volatile("v_add_u32 %1, %0, %2\n v_add_u32 %0, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
        if (MODE == 21) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_lshrrev_b32 %0, 3, %0\n v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(b));) }
        if (MODE == 22) { REP64(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 23) { REP64(asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));) }
        if (MODE == 24) { REP64(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(m));) }
        if (MODE == 25) { REP64(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(m));) }
        if (MODE == 26) { REP64(asm volatile("v_add_u32 %0, %0, %1\n s_nop 3" : "+v"(a) : "v"(m));) }
        if (MODE == 27) { REP64(asm volatile("v_cmp_le_u32 vcc, %0, %1\n v_bcnt_u32_b32 %0, vcc_lo, %0" : "+v"(a) : "v"(m) : "vcc");) }
        if (MODE == 28) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n s_nop 3\n v_readlane_b32 s21, %1, s20\n v_add_u32 %0, s21, %0" : "+v"(a) : "v"(b) : "s20", "s21");) a &= 63; }
        if (MODE == 29) { REP64(asm volatile("ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b));) a &= 0xfc; }
        if (MODE == 30) { REP64(asm volatile("s_cmp_ge_u32 s20, s21\n s_cselect_b32 s20, s22, s23" ::: "s20", "scc");) }
        if (MODE == 31) { REP64(asm volatile("s_mov_b32 m0, s20\n s_nop 0\n s_movrels_b32 s20, s24" ::: "s20", "m0");) }
        if (MODE == 32) { REP64(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        if (MODE == 33) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n s_lshr_b32 s21, s20, 3\n s_and_b32 s21, s21, 15\n s_add_u32 s20, s20, s21\n v_mov_b32 %0, s20" : "+v"(a) :: "s20", "s21", "scc");) }
        if (MODE == 34) { REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %1" : "+v"(a) : "v"(m) : "s20");) }
        if (MODE == 35) { REP64(asm volatile("s_add_u32 s20, s20, 3\n v_add_u32 %0, s20, %0" : "+v"(a) :: "s20", "scc");) }
        if (MODE == 36) { REP64(asm volatile("v_cmp_le_u32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(m) : "vcc", "s20", "s21", "scc");) }
        if (MODE == 37) { REP64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xf0, %2" : "=v"(q4), "+v"(a) : "v"(q4.x));) }
        if (MODE == 38) { REP64(asm volatile("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)" : "+v"(a) : "s"(out));) a &= 0xfc; }
        if (MODE == 40) { REP64(asm volatile("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n 1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m) : "scc");) }
        if (MODE == 41) { REP64(asm volatile("s_cmp_lg_u32 s20, s20\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n 1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m) : "scc");) }
        if (MODE == 42) { REP64(asm volatile("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n 1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m) : "scc");) }
        if (MODE == 43) { REP64(asm volatile("s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 44) { REP64(asm volatile("s_waitcnt vmcnt(0)\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
        if (MODE == 45) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_ff1_i32_b64 s20, vcc\n v_readlane_b32 s21, %0, s20\n v_add_u32 %0, s21, %0" : "+v"(a) : "v"(m) : "vcc", "s20", "s21");) }
        if (MODE == 46) { REP64(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n 1:\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m) : "vcc");) }
        if (MODE == 47) { REP64(asm volatile("s_lshr_b64 s[20:21], s[20:21], 8\n s_add_u32 s22, s22, 1" ::: "s20", "s21", "s22", "scc");) }
        if (MODE == 50) { REP64(asm volatile("v_mov_b32_dpp %0, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %1, 8, %0\n v_cmp_ge_u32_e64 s[22:23], %1, %8\n v_cmp_ge_u32_e64 s[20:21], %0, %8\n v_lshrrev_b32 %2, 16, %0\n v_cndmask_b32_e64 %3, %1, %2, s[22:23]\n v_cndmask_b32_e64 %4, %0, %3, s[20:21]\n v_mul_hi_u32 %5, %4, %9\n v_lshrrev_b32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32 %6, %4, %8\n v_mad_u32_u24 %7, %5, %10, %6\n v_cndmask_b32_e64 %11, %11, %0, s[24:25]\n s_nop 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(m) :: "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (MODE == 51) { REP64(asm volatile("v_mov_b32 %0, %7\n v_lshrrev_b32 %1, 8, %0\n v_cmp_ge_u32_e64 s[22:23], %1, %8\n v_cmp_ge_u32_e64 s[20:21], %0, %8\n v_lshrrev_b32 %2, 16, %0\n v_cndmask_b32_e64 %3, %1, %2, s[22:23]\n v_cndmask_b32_e64 %4, %0, %3, s[20:21]\n v_mul_hi_u32 %5, %4, %9\n v_lshrrev_b32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32 %6, %4, %8\n v_mad_u32_u24 %7, %5, %10, %6\n v_cndmask_b32_e64 %11, %11, %0, s[24:25]\n s_nop 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(m) :: "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (MODE == 52) { REP64(asm volatile("v_mov_b32_dpp %0, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %1, 8, %0\n v_cmp_ge_u32_e64 s[22:23], %1, %8\n v_cmp_ge_u32_e64 s[20:21], %0, %8\n v_lshrrev_b32 %2, 16, %0\n v_cndmask_b32_e64 %3, %1, %2, s[22:23]\n v_cndmask_b32_e64 %4, %0, %3, s[20:21]\n v_mul_hi_u32 %5, %4, %9\n v_lshrrev_b32 %5, 3, %5\n v_add_u32 %6, %4, %8\n v_mad_u32_u24 %7, %5, %10, %6\n v_cndmask_b32_e64 %11, %11, %0, s[24:25]\n s_nop 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(m) :: "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (MODE == 53) { REP64(asm volatile("v_mov_b32_dpp %0, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %1, 8, %0\n v_cmp_ge_u32_e64 s[22:23], %1, %8\n v_cmp_ge_u32_e64 s[20:21], %0, %8\n v_lshrrev_b32 %2, 16, %0\n v_cndmask_b32_e64 %3, %1, %2, s[22:23]\n v_cndmask_b32_e64 %4, %0, %3, s[20:21]\n v_add_u32 %5, %4, %9\n v_lshrrev_b32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32 %6, %4, %8\n v_mad_u32_u24 %7, %5, %10, %6\n v_cndmask_b32_e64 %11, %11, %0, s[24:25]\n s_nop 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(m) :: "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (MODE == 54) { REP64(asm volatile("v_mov_b32_dpp %0, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %1, 8, %0\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %0, %8\n v_lshrrev_b32 %2, 16, %0\n v_xor_b32 %3, %1, %2\n v_xor_b32 %4, %0, %3\n v_mul_hi_u32 %5, %4, %9\n v_lshrrev_b32_sdwa %5, %10, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_add_u32 %6, %4, %8\n v_mad_u32_u24 %7, %5, %10, %6\n v_xor_b32 %11, %11, %0\n s_nop 0" : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(m) :: "s20", "s21", "s22", "s23", "s24", "s25");) }
        if (MODE == 17) { REP64(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_or_b32 %0, %0, %1" : "+v"(a) : "v"(m));) }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a + b + c + d + q4.x + w0 + w1 + w2 + w3 + w4 + w5 + w6;
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE> void run(const char *name, int per_rep, uint32_t *o, uint64_t *clk)
{
    uint64_t h[2];
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, o, iters, clk, 12345u);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ns = (double)h[1] * 10.0;                       // memrealtime: 100 MHz
    const double n = (double)iters * 64 * per_rep;
    printf("%-44s %6.2f ns/instr  (memtime ticks/instr %.2f)\n", name, ns / n, (double)h[0] / n);
}

int main()
{
    uint32_t *o; uint64_t *clk;
    hipMalloc(&o, 1024); hipMalloc(&clk, 16);
    run<0>("v_add_u32 dependent", 1, o, clk);
    run<1>("v_add_u32 x4 independent (per instr)", 4, o, clk);
    run<2>("v_mul_hi_u32 dependent", 1, o, clk);
    run<6>("v_mul_hi_u32 x4 independent (per instr)", 4, o, clk);
    run<3>("v_mul_lo_u32 dependent", 1, o, clk);
    run<4>("v_mul_u32_u24 dependent", 1, o, clk);
    run<11>("v_mad_u32_u24 dependent", 1, o, clk);
    run<12>("v_add3_u32 dependent", 1, o, clk);
    run<5>("v_cmp+v_cndmask dependent (per pair)", 1, o, clk);
    run<17>("v_lshrrev+v_or dependent (per pair)", 1, o, clk);
    run<22>("v_xor dependent", 1, o, clk);
    run<23>("v_lshrrev dependent", 1, o, clk);
    run<18>("add,xor dependent (per pair)", 1, o, clk);
    run<19>("add,xor,lshr,or dependent (per quad)", 1, o, clk);
    run<20>("add a->b, add b->a (per pair)", 1, o, clk);
    run<25>("two independent add chains interleaved (per pair)", 1, o, clk);
    run<24>("add dependent + s_nop 0 (per pair)", 1, o, clk);
    run<21>("mul_hi,lshr,mad_u24 dependent (per triple)", 1, o, clk);
    run<26>("add + s_nop 3 (per pair)", 1, o, clk);
    run<27>("v_cmp vcc + v_bcnt(vcc_lo) (per pair)", 1, o, clk);
    run<28>("readfirstlane,nop3,readlane(sgpr sel),v_add (quad)", 1, o, clk);
    run<29>("ds_bpermute dependent + wait", 1, o, clk);
    run<30>("s_cmp + s_cselect (per pair)", 1, o, clk);
    run<31>("s_mov m0, nop, s_movrels (per triple)", 1, o, clk);
    run<32>("3x v_add_dpp row_shr (per triple)", 1, o, clk);
    run<33>("readfirstlane + 3 salu + v_mov (5)", 1, o, clk);
    run<34>("readfirstlane + v_add(sgpr) (per pair)", 1, o, clk);
    run<35>("s_add + v_add(sgpr) (per pair)", 1, o, clk);
    run<36>("v_cmp, s_and(vcc), v_cndmask (triple)", 1, o, clk);
    run<37>("ds_read_b128 dependent + wait + and", 1, o, clk);
    run<38>("global_load_dword dependent (L2/L1 hit)", 1, o, clk);
    run<40>("s_cmp + TAKEN s_cbranch skip 1 + v_add (3 issued)", 1, o, clk);
    run<41>("s_cmp + NOT taken s_cbranch + 2 v_add (4 issued)", 1, o, clk);
    run<42>("s_cmp + TAKEN s_cbranch skip 20 + v_add (3 issued)", 1, o, clk);
    run<43>("s_waitcnt lgkmcnt(0) idle + v_add", 1, o, clk);
    run<44>("s_waitcnt vmcnt(0) idle + v_add", 1, o, clk);
    run<45>("v_cmp, s_ff1(vcc), v_readlane(sel), v_add (quad)", 1, o, clk);
    run<46>("v_cmp vcc + s_cbranch_vccz not taken + 2 v_add", 1, o, clk);
    run<47>("s_lshr_b64 + s_add (pair)", 1, o, clk);
    run<50>("rANS encoder step, as in k_rans_lanes (13 slots)", 1, o, clk);
    run<51>("  same, plain v_mov instead of the DPP move", 1, o, clk);
    run<52>("  same, plain shift instead of the SDWA shift", 1, o, clk);
    run<53>("  same, v_add instead of v_mul_hi_u32", 1, o, clk);
    run<54>("  same, no SGPR masks (add/xor instead of cmp/cndmask)", 1, o, clk);
    run<7>("v_readlane+v_add dependent (per pair)", 1, o, clk);
    run<15>("readfirstlane+s_add+v_mov (per triple)", 1, o, clk);
    run<10>("v_cmp+s_bcnt1+v_add (per triple)", 1, o, clk);
    run<9>("dpp wave_shl+v_add (per pair)", 1, o, clk);
    run<8>("ds_read_b32 dependent + wait", 1, o, clk);
    run<16>("ds_write_b32 + v_add (per pair)", 1, o, clk);
    run<13>("s_add_u32 dependent", 1, o, clk);
    run<14>("s_mul_hi_u32 dependent", 1, o, clk);
    return 0;
}
