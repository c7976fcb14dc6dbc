"""Writes halo2-pse_amd/csrc/fieldu_chain.inc: N multiply-adds into one 64-bit accumulator as ONE asm statement, N = 1..9."""
import os

def chain(name, instr, t0, c0, t1, c1):
    out = ['template <int N>\n__device__ __forceinline__ void %s(int64_t& acc, const %s (&x)[N], const %s (&y)[N]) {\n    uint64_t sink;\n' % (name, t0, t1)]
    for n in range(1, 10):
        body = '\\n\\t'.join('%s %%0, %%1, %%%d, %%%d, %%0' % (instr, 2 + 2 * i, 3 + 2 * i) for i in range(n))
        ops = ', '.join('"%s"(x[%d]), "%s"(y[%d])' % (c0, i, c1, i) for i in range(n))
        out.append('    %s constexpr (N == %d)\n        asm("%s" : "+v"(acc), "=&s"(sink) : %s);\n' % ('if' if n == 1 else 'else if', n, body, ops))
    out.append('}\n')
    return ''.join(out)

HEADER = """// fieldu_chain.inc -- written by tools/gen_fieldu_chain.py, do not edit.
// N multiply-adds into one 64-bit accumulator as ONE asm statement.  The AMDGPU hazard recognizer puts an s_nop after every
// inline asm whose result the next instruction reads (it cannot see that v_mad_i64_i32 needs none; compiler-generated
// chains get none), so one statement per multiply-add cost ~170 s_nop per field multiplication: hidden at 4 busy waves
// per SIMD, 1.5x at one.  The carry-out register is early-clobber: it must not share a register with a later input.
"""
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "..", "halo2-pse_amd", "csrc", "fieldu_chain.inc"), "w") as f:
    f.write(HEADER + chain("fu_chain_ss", "v_mad_i64_i32", "int32_t", "v", "int32_t", "v") + "\n" + chain("fu_chain_mp", "v_mad_u64_u32", "uint32_t", "v", "uint32_t", "s"))
