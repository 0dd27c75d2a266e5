#!/usr/bin/env python3
"""Generates tools/build/valu_tput2.hip: per-instruction THROUGHPUT probes for gfx950 beyond tools/valu_tput.hip (selects,
compares, integer ops, transcendentals, lane reads, LDS writes).  Each kernel fills every SIMD with 8 one-wave workgroups
issuing 16 independent copies of one instruction per loop trip; wall time -> ns per wave-instruction per SIMD, printed
relative to v_fmac_f64 (4 cycles per wave64 by the fp64 peak).
usage: python3 tools/gen_valu_tput2.py && hipcc -O3 --offload-arch=gfx950 tools/build/valu_tput2.hip -o tools/build/valu_tput2"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, asm template with {i}-indexed operands, operand list)
# operand classes: A = float a[i] (+v), Ao = float a[i] (=v), D = double d[i] (+v), Do = double d[i] (=v), x,y float in, dx,dy double in,
# S = sgpr out (=s) si[i], M = 64-bit sgpr mask in (s) m64, U = unsigned u[i] (+v), Uo
P = [
 ("v_fmac_f64", "v_fmac_f64 %0, %1, %2", ["D", "dx", "dy"]),
 ("v_fmac_f32", "v_fmac_f32 %0, %1, %2", ["A", "x", "y"]),
 ("v_cndmask_b32 vcc", "v_cndmask_b32 %0, %1, %2, vcc", ["Ao", "x", "y"]),
 ("v_cndmask_b32_e64 sgpr", "v_cndmask_b32_e64 %0, %1, %2, %3", ["Ao", "x", "y", "M"]),
 ("v_cmp_lt_f32 vcc", "v_cmp_lt_f32 vcc, %0, %1", ["ain", "y"]),
 ("v_cmp_lt_f64 vcc", "v_cmp_lt_f64 vcc, %0, %1", ["din", "dy"]),
 ("v_cmp_lt_f32_e64 sgpr", "v_cmp_lt_f32_e64 %0, %1, %2", ["S64", "ain", "y"]),
 ("v_cmp_eq_u32 vcc", "v_cmp_eq_u32 vcc, %0, %1", ["uin", "ux"]),
 ("cmp+cndmask pair", "v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc", ["Ao", "x", "y"]),
 ("v_and_b32", "v_and_b32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %1", ["Uo", "ux"]),
 ("v_add_u32", "v_add_u32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_mad_u32_u24", "v_mad_u32_u24 %0, %1, %2, %1", ["Uo", "ux", "uy"]),
 ("v_mul_lo_u32", "v_mul_lo_u32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_bfe_u32", "v_bfe_u32 %0, %1, 4, 4", ["Uo", "ux"]),
 ("v_xor_b32", "v_xor_b32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_readlane_b32", "v_readlane_b32 %0, %1, 3", ["S", "x"]),
 ("v_readfirstlane_b32", "v_readfirstlane_b32 %0, %1", ["S", "x"]),
 ("v_rsq_f64", "v_rsq_f64 %0, %1", ["Do", "dx"]),
 ("v_sqrt_f64", "v_sqrt_f64 %0, %1", ["Do", "dx"]),
 ("v_rsq_f32", "v_rsq_f32 %0, %1", ["Ao", "x"]),
 ("v_sqrt_f32", "v_sqrt_f32 %0, %1", ["Ao", "x"]),
 ("v_mul_f32", "v_mul_f32 %0, %1, %2", ["Ao", "x", "y"]),
 ("v_add_f32", "v_add_f32 %0, %1, %2", ["Ao", "x", "y"]),
 ("v_fma_f64 (3 src)", "v_fma_f64 %0, %1, %2, %3", ["Do", "dx", "dy", "dz"]),
 ("v_fma_f64 neg/abs mods", "v_fma_f64 %0, -%1, |%2|, %3", ["Do", "dx", "dy", "dz"]),
 ("v_max_f64", "v_max_f64 %0, %1, %2", ["Do", "dx", "dy"]),
 ("v_ldexp_f64", "v_ldexp_f64 %0, %1, %2", ["Do", "dx", "ux"]),
 ("v_trunc_f64", "v_trunc_f64 %0, %1", ["Do", "dx"]),
 ("v_rndne_f64", "v_rndne_f64 %0, %1", ["Do", "dx"]),
 ("v_cvt_i32_f64", "v_cvt_i32_f64 %0, %1", ["Uo", "dx"]),
 ("v_cvt_f64_i32", "v_cvt_f64_i32 %0, %1", ["Do", "ux"]),
 ("v_cvt_f32_f64", "v_cvt_f32_f64 %0, %1", ["Ao", "dx"]),
 ("v_cvt_f64_f32", "v_cvt_f64_f32 %0, %1", ["Do", "x"]),
 ("v_div_scale_f64", "v_div_scale_f64 %0, vcc, %1, %2, %1", ["Do", "dx", "dy"]),
 ("v_div_fmas_f64", "v_div_fmas_f64 %0, %1, %2, %3", ["Do", "dx", "dy", "dz"]),
 ("v_div_fixup_f64", "v_div_fixup_f64 %0, %1, %2, %3", ["Do", "dx", "dy", "dz"]),
 ("v_fmac_f64 sgpr src", "v_fmac_f64 %0, %1, %2", ["D", "SD", "dy"]),
 ("v_mov_b32 sgpr src", "v_mov_b32 %0, %1", ["Ao", "SF"]),
 ("v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", ["Ao", "x"]),
 ("v_mov_b32_dpp bank_mask", "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0x5", ["A", "x"]),
 ("v_accvgpr_write_b32", "v_accvgpr_write_b32 a{i}, %0", ["x"]),
 ("v_accvgpr_read_b32", "v_accvgpr_read_b32 %0, a{i}", ["Ao"]),
 ("ds_write_b32", "ds_write_b32 %0, %1", ["addr", "x"]),
 ("ds_write_b64", "ds_write_b64 %0, %1", ["addr", "dx"]),
 ("ds_write_b128", "ds_write_b128 %0, %1", ["addr", "q4"]),
 ("ds_read_b32 per-lane", "ds_read_b32 %0, %1", ["Ao", "laddr"]),
 ("ds_read_b64 per-lane", "ds_read_b64 %0, %1", ["Do", "laddr8"]),
 ("ds_read_b128 per-lane", "ds_read_b128 %0, %1", ["Q", "laddr16"]),
 ("s_and_b64 (SALU)", "s_and_b64 %0, %1, %1", ["S64", "M"]),
 ("v_sin_f32", "v_sin_f32 %0, %1", ["Ao", "x"]),
 ("v_exp_f32", "v_exp_f32 %0, %1", ["Ao", "x"]),
 ("v_rcp_f32", "v_rcp_f32 %0, %1", ["Ao", "x"]),
 ("v_rcp_f64", "v_rcp_f64 %0, %1", ["Do", "dx"]),
 ("v_pk_add_f32", "v_pk_add_f32 %0, %1, %2", ["Do", "dx", "dy"]),
 ("v_swap_b32", "v_swap_b32 %0, %1", ["A", "A2"]),
]
P2 = [
 ("v_fmac_f64", "v_fmac_f64 %0, %1, %2", ["D", "dx", "dy"]),
 ("v_fmac_f32", "v_fmac_f32 %0, %1, %2", ["A", "x", "y"]),
 ("v_fma_f32 sgpr src", "v_fma_f32 %0, %1, %2, %0", ["A", "SF", "y"]),
 ("v_fma_f32 inline 0.5", "v_fma_f32 %0, %1, 0.5, %0", ["A", "x"]),
 ("v_mul_f32 inline 2.0", "v_mul_f32 %0, 2.0, %1", ["Ao", "x"]),
 ("v_mul_f32 literal", "v_mul_f32 %0, 0x3fc90fdb, %1", ["Ao", "x"]),
 ("v_mul_f32 sgpr", "v_mul_f32 %0, %1, %2", ["Ao", "SF", "x"]),
 ("v_cndmask_b32_e32 vcc (no clobber)", "v_cndmask_b32_e32 %0, %1, %2, vcc", ["Ao", "x", "y"]),
 ("v_cndmask_b32_e64 vcc", "v_cndmask_b32_e64 %0, %1, %2, vcc", ["Ao", "x", "y"]),
 ("v_cndmask_b32_e64 sgpr", "v_cndmask_b32_e64 %0, %1, %2, %3", ["Ao", "x", "y", "M"]),
 ("v_cndmask_b32_e64 0,v,sgpr", "v_cndmask_b32_e64 %0, 0, %1, %2", ["Ao", "x", "M"]),
 ("v_add_u32 inline", "v_add_u32 %0, 16, %1", ["Uo", "ux"]),
 ("v_add_u32 sgpr", "v_add_u32 %0, %1, %2", ["Uo", "SF", "ux"]),
 ("v_mov_b32 inline", "v_mov_b32 %0, 1.0", ["Ao"]),
 ("v_lshlrev_b32 vgpr amt", "v_lshlrev_b32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_lshl_add_u32", "v_lshl_add_u32 %0, %1, 2, %2", ["Uo", "ux", "uy"]),
 ("v_add3_u32", "v_add3_u32 %0, %1, %2, %1", ["Uo", "ux", "uy"]),
 ("v_and_or_b32", "v_and_or_b32 %0, %1, %2, %1", ["Uo", "ux", "uy"]),
 ("v_or_b32", "v_or_b32 %0, %1, %2", ["Uo", "ux", "uy"]),
 ("v_sub_f32", "v_sub_f32 %0, %1, %2", ["Ao", "x", "y"]),
 ("v_max_f32", "v_max_f32 %0, %1, %2", ["Ao", "x", "y"]),
 ("v_mul_f32 neg mod (e64)", "v_mul_f32_e64 %0, -%1, %2", ["Ao", "x", "y"]),
 ("v_fma_f32 neg mod", "v_fma_f32 %0, -%1, %2, %0", ["A", "x", "y"]),
 ("v_cmp_lt_f32_e32 vcc", "v_cmp_lt_f32_e32 vcc, %0, %1", ["ain", "y"]),
 ("v_cmp_class_f32", "v_cmp_class_f32_e32 vcc, %0, %1", ["ain", "ux"]),
 ("v_mov_b64_dpp newbcast", "v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf", ["Do", "dx"]),
 ("v_mov_b32_e32", "v_mov_b32_e32 %0, %1", ["Ao", "x"]),
 ("v_pk_mov_b32", "v_pk_mov_b32 %0, %1, %2", ["Do", "dx", "dy"]),
 ("v_fmac_f32 x2 dependent", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2", ["A", "x", "y"]),
 ("v_fmac_f64 x2 dependent", "v_fmac_f64 %0, %1, %2\n v_fmac_f64 %0, %1, %2", ["D", "dx", "dy"]),
 ("v_rsq_f64 + dep fma", "v_rsq_f64 %0, %1\n v_fma_f64 %0, %0, %2, %0", ["Do", "dx", "dy"]),
]
import sys
if len(sys.argv) > 1 and sys.argv[1] == "2":
    P = P2
OPS = {
 "A": '"+v"(a[{i}])', "Ao": '"=v"(a[{i}])', "A2": '"+v"(a2[{i}])', "D": '"+v"(d[{i}])', "Do": '"=v"(d[{i}])', "U": '"+v"(u[{i}])',
 "Uo": '"=v"(u[{i}])', "S": '"=s"(si[{i}])', "S64": '"=s"(sl[{i}])', "Q": '"=v"(q[{i} & 3])',
 "x": '"v"(x)', "y": '"v"(y)', "dx": '"v"(dx)', "dy": '"v"(dy)', "dz": '"v"(dz)', "ux": '"v"(ux)', "uy": '"v"(uy)',
 "ain": '"v"(a[{i}])', "din": '"v"(d[{i}])', "uin": '"v"(u[{i}])', "M": '"s"(m64)', "SD": '"s"(sd)', "SF": '"s"(sf)',
 "addr": '"v"(waddr)', "laddr": '"v"(laddr)', "laddr8": '"v"(laddr8)', "laddr16": '"v"(laddr16)', "q4": '"v"(qin)',
}
OUT = {"A", "Ao", "A2", "D", "Do", "U", "Uo", "S", "S64", "Q"}

def stmt(tmpl, ops, i):
    outs = [OPS[o].format(i=i) for o in ops if o in OUT]
    ins = [OPS[o].format(i=i) for o in ops if o not in OUT]
    # renumber: outputs first then inputs in template order -> build mapping
    order = [o for o in ops if o in OUT] + [o for o in ops if o not in OUT]
    t = tmpl.replace("{i}", str(i))
    # template numbers refer to ops order; remap to constraint order
    idx = {}
    used = [False] * len(order)
    for k, o in enumerate(ops):
        for j, oo in enumerate(order):
            if oo == o and not used[j]:
                used[j] = True; idx[k] = j; break
    for k in sorted(idx, reverse=True):
        t = t.replace("%%%d" % k, "%%<%d>" % idx[k])
    t = t.replace("%<", "%").replace(">", "")
    clob = ' : "vcc"' if ("vcc" in tmpl and "cndmask" not in tmpl) else ""
    if tmpl.startswith("v_accvgpr_write"):
        clob = ' : "a%d"' % i
    return '            asm volatile("%s" : %s : %s%s);' % (t.replace("\n", "\\n"), ", ".join(outs), ", ".join(ins), clob)

src = ['// generated by tools/gen_valu_tput2.py', '#include <hip/hip_runtime.h>', '#include <cstdio>', 'constexpr int ITERS = 2048;',
       'typedef float f4 __attribute__((ext_vector_type(4)));']
for m, (name, tmpl, ops) in enumerate(P):
    lgkm = tmpl.startswith("ds_")
    src.append('__global__ __launch_bounds__(64) void probe%d(float* sink) {' % m)
    src.append('    float a[16], a2[16]; double d[16]; unsigned u[16]; unsigned si[16]; unsigned long long sl[16]; f4 q[4]; __shared__ float lds[2048];')
    src.append('    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; a2[i] = i; d[i] = threadIdx.x + i; u[i] = threadIdx.x * i; si[i] = 0; sl[i] = 0; }')
    src.append('    for (int i = 0; i < 4; ++i) q[i] = f4{0, 0, 0, 0};')
    src.append('    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = i;')
    src.append('    float x = 1.0001f + threadIdx.x, y = 0.5f; double dx = 1.0001 + threadIdx.x, dy = 0.5, dz = 0.25; unsigned ux = threadIdx.x, uy = 77;')
    src.append('    unsigned long long m64 = 0x5555555555555555ull ^ (unsigned long long)sink; double sd = 1.5; float sf = 2.5f; f4 qin = {1, 2, 3, 4};')
    src.append('    asm volatile("" : "+s"(m64), "+s"(sd), "+s"(sf));')
    src.append('    unsigned waddr = threadIdx.x * 16, laddr = threadIdx.x * 4, laddr8 = threadIdx.x * 8, laddr16 = threadIdx.x * 16;')
    src.append('#pragma unroll 1')
    src.append('    for (int it = 0; it < ITERS; ++it) {')
    for i in range(16):
        src.append(stmt(tmpl, ops, i))
    if lgkm:
        src.append('            asm volatile("s_waitcnt lgkmcnt(0)");')
    src.append('    }')
    src.append('    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + a2[i] + float(d[i]) + float(u[i]) + float(si[i]) + float(sl[i]);')
    src.append('    s += q[0].x + q[1].y + q[2].z + q[3].w;')
    src.append('    if (s == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];')
    src.append('}')
src.append('typedef void (*kern_t)(float*);')
src.append('int main() {')
src.append('    float* sink; if (hipMalloc(&sink, 4 * 64 * 16384) != hipSuccess) return 1;')
src.append('    kern_t ks[] = {%s};' % ", ".join("probe%d" % m for m in range(len(P))))
src.append('    const char* names[] = {%s};' % ", ".join('"%s"' % p[0] for p in P))
src.append('    const int W = 8, blocks = 1024 * W; float ref = 0;')
src.append('    for (int m = 0; m < %d; ++m) {' % len(P))
src.append('        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);')
src.append('        ks[m]<<<blocks, 64>>>(sink); (void)hipDeviceSynchronize();')
src.append('        (void)hipEventRecord(e0); ks[m]<<<blocks, 64>>>(sink); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);')
src.append('        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); if (m == 0) ref = ms;')
src.append('        printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD  = %5.2f cycles (v_fmac_f64 = 4)\\n", names[m], ms, ms * 1e6 / (double(ITERS) * 16 * W), 4.0 * ms / ref);')
src.append('        fflush(stdout);')
src.append('    }')
src.append('    return 0;')
src.append('}')
os.makedirs(os.path.join(ROOT, "tools", "build"), exist_ok=True)
open(os.path.join(ROOT, "tools", "build", "valu_tput%s.hip" % (sys.argv[1] if len(sys.argv) > 1 else "2a")), "w").write("\n".join(src) + "\n")
print("wrote %d probes" % len(P))
