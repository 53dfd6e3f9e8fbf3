/*
 * hh_policy_kernel_w.h — what the weights-through-LDS forms of the policy forward (hh_policy_kernel_w16.h) share: the 1 KB MFMA A-fragment
 * ("piece") of the per-network weight stream, its LDS-DMA copy (global_load_lds_dwordx4: 1 KB per wave-instruction, no staging registers)
 * and its conflict-free ds_read_b128 fetch.
 *
 * Round 6 retired the form this header was written for — hh_k_policy_w: 32 rows per wave on v_mfma_f32_32x32x16_f16, 256 registers of hidden
 * row, ONE wave per SIMD, 128 rows per workgroup — together with the fp32-MFMA forward hh_k_policy: no row count selected either of them
 * (46.7 .. 120.6 us per call against hh_k_policy_w16's 30.8 .. 117.4 over 4096 .. 65536 rows; the fp32 form 2.4x
 * slower than the split-fp16 tile form at equal accuracy), they were reachable through HH_POLICY_W=1 / HH_POLICY_FP32=1 only, and every
 * change to the shared pieces had to be soaked through them.  What that form taught (a wave's own epilogue does not hide behind its own
 * MFMAs at one wave per SIMD) is recorded in DESIGN.md; git history holds the code.
 */
#ifndef HH_POLICY_KERNEL_W_H
#define HH_POLICY_KERNEL_W_H

#define HHW_PIECE 1024 /* bytes of one fragment: 64 lanes x 8 halves */

typedef __attribute__((address_space(3))) unsigned char hhw_lds_u8;
typedef const __attribute__((address_space(1))) unsigned char hhw_glb_u8;

/* one 1 KB piece: global (per lane) -> LDS (wave-uniform base), both addresses + o x 1 KB through the instruction's immediate offset */
__device__ __forceinline__ void hhw_glds(const unsigned char *s, unsigned char *d, int o) {
    switch (o) { /* the builtin wants a literal; o is a constant after unrolling */
    case 0: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 0, 0); break;
    case 1: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, HHW_PIECE, 0); break;
    case 2: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 2 * HHW_PIECE, 0); break;
    default: __builtin_amdgcn_global_load_lds((hhw_glb_u8 *)s, (hhw_lds_u8 *)d, 16, 3 * HHW_PIECE, 0); break;
    }
}

__device__ __forceinline__ hh_h8 hhw_frag(const unsigned char *lbuf, int piece, int lane) {
    return hhp_as_h8(*reinterpret_cast<const float4 *>(lbuf + piece * HHW_PIECE + lane * 16));
}


/* "these fragments are needed now": pins the LDS reads' completion in front of the MFMAs that follow (hipcc otherwise sinks each read to its use) */
__device__ __forceinline__ void hhw_need4(hh_h8 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }

/* pieces first .. first + n - 1 of this wave's (contiguous) share of a chunk; s / d = the lane's source byte / the LDS address of the share's first piece */
__device__ __forceinline__ void hhw_issue_some(const unsigned char *__restrict__ s, unsigned char *d, int first, int n) {
#pragma unroll
    for (int u = 0; u < n; u++) {
        const int i = first + u;
        hhw_glds(s + (size_t)(i >> 2) * 4 * HHW_PIECE, d + (i >> 2) * 4 * HHW_PIECE, i & 3);
    }
}


#endif /* HH_POLICY_KERNEL_W_H */
