# generate a HIP ubench with fixed instruction orders (inline asm) to probe add/pk_min run-length effects
pats = {}
def A(d, s0, s1): return f"v_add_u32 %{d}, %{s0}, %{s1}"
def M(d, s0, s1): return f"v_pk_min_u16 %{d}, %{s0}, %{s1}"
# operands: 0..7 = a[0..7] (add chains), 8..15 = m[0..7] (min chains), 16,17 = constants k1,k2
def runs(na, nm, dep=False):
    out = []; ai = 0; mi = 0
    total_a, total_m = 32, 16
    while ai < total_a or mi < total_m:
        for _ in range(na):
            if ai < total_a:
                i = ai % 8; out.append(A(i, i, 16 + (ai & 1))); ai += 1
        for _ in range(nm):
            if mi < total_m:
                i = mi % 8
                if dep: out.append(M(8 + i, i, (i + 1) % 8))   # consumes fresh add results
                else: out.append(M(8 + i, 8 + i, 8 + (i + 3) % 8))
                mi += 1
    return out
pats["A2M1 indep"] = runs(2, 1)
pats["A4M2 indep"] = runs(4, 2)
pats["A8M4 indep"] = runs(8, 4)
pats["A16M8 indep"] = runs(16, 8)
pats["A32M16 indep"] = runs(32, 16)
pats["A2M1 dep"] = runs(2, 1, True)
pats["A16M8 dep"] = runs(16, 8, True)
pats["A only"] = [A(i % 8, i % 8, 16 + (i & 1)) for i in range(48)]
pats["M only"] = [M(8 + i % 8, 8 + i % 8, 8 + (i + 3) % 8) for i in range(48)]
src = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdint.h>',
       'template <int P> __global__ void k(uint32_t *out, uint32_t seed, int iters, uint32_t s1, uint32_t s2) {',
       ' uint32_t a[8], m[8]; for (int i = 0; i < 8; i++) { a[i] = seed * (i + 1) + threadIdx.x; m[i] = a[i] ^ 0x1234567; }',
       ' uint32_t k1 = s1 + (threadIdx.x & 1), k2 = s2 + (threadIdx.x & 2);',
       ' for (int it = 0; it < iters; it++) {']
names = list(pats)
for pi, n in enumerate(names):
    body = "\\n\\t".join(pats[n])
    ops = ", ".join([f'"+v"(a[{i}])' for i in range(8)] + [f'"+v"(m[{i}])' for i in range(8)])
    src.append(f'  if (P == {pi}) asm volatile("{body}" : {ops} : "v"(k1), "v"(k2));')
src += [' }', ' uint32_t acc = 0; for (int i = 0; i < 8; i++) acc ^= a[i] ^ m[i]; out[blockIdx.x * 64 + threadIdx.x] = acc; }',
        'template <int P> static void run(const char *name, int n, int w, uint32_t *d) {',
        ' const int iters = 2000, blocks = 256 * 4 * w; hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);',
        ' hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(64), 0, 0, d, 12345u, 10, 3u, 5u); (void)hipDeviceSynchronize();',
        ' (void)hipEventRecord(a); hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(64), 0, 0, d, 12345u, iters, 3u, 5u); (void)hipEventRecord(b); (void)hipEventSynchronize(b);',
        ' float ms; (void)hipEventElapsedTime(&ms, a, b);',
        ' printf("%-16s waves/SIMD=%d  %.2f cyc/instr  (%.1f cyc per 48-instr group)\\n", name, w, ms * 1e6 / ((double)iters * n * w) * 2.4, ms * 1e6 / ((double)iters * w) * 2.4); }',
        'int main() { uint32_t *d; (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4); for (int w : {2, 4, 8}) {']
for pi, n in enumerate(names):
    src.append(f'  run<{pi}>("{n}", {len(pats[n])}, w, d);')
src += [' } return 0; }']
open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'valu_runs.hip'), 'w').write("\n".join(src))
