// rgx.hpp -- what the chaining tiers (k_regions.hip) export per strand search for the launches that turn chains into regions:
// k_x4prep + k_ext4 (k_ext4.hip: the extensions of every chain's best seed, four to a wavefront) and k_c2r (k_regions.hip: the
// reference's seed loop, mem_chain2region1, memchain.c:742-870, which takes those extensions from the record).
// A record = RgXHdr | RgXChain[n_chains] | RgXSeed[n_seeds] | RgXExt[n_chains] (the last only when hdr.has_ext), at xoff[t] in the pool.
#pragma once
#include <stdint.h>
#include "kernels.h"

struct RgXHdr { int n_chains, n_seeds; float frac_rep; int flt; int has_ext, pad; };   // flt: min_HSP_score when the seed-SW filter applies to this read (k_seedsw runs before k_c2r), RG_NOFLT otherwise
#define RG_NOFLT ((int)0x80000000)
struct RgXChain { long long pos; int rid, seed_off; unsigned short n_main, n_extra; int pad; };
struct RgXSeed { long long rbeg; short qbeg, len; int sb; };   // sb: mem_seed_t.score << 1 | failed asymmetric_flt_seed
#define XS_BAD(x) ((x).sb & 1)
#define XS_SCORE(x) ((x).sb >> 1)
// The two extensions (memchain.c:613-730) of the seed mem_chain2region1's best-first loop reaches first in a chain's main list, made
// ahead of that loop: what it needs of them to fill in the region.  status: 0 none (k_c2r extends inline), 1 valid
struct RgXExt { long long rb, re; int qb, qe, score, truesc; int aw0, aw1; int si, status; };
struct RgXPool { unsigned char *base; unsigned long long cap; unsigned long long *cursor; long long *xoff; int *xlist; unsigned int *xcount; int ext; };
static inline RgXPool rgx_pool(const RgXPoolArg *XA)
{
	RgXPool X; X.base = nullptr; X.cap = 0; X.cursor = nullptr; X.xoff = nullptr; X.xlist = nullptr; X.xcount = nullptr; X.ext = 0;
	if (XA) { X.base = XA->base; X.cap = XA->cap; X.cursor = XA->cursor; X.xoff = XA->xoff; X.xlist = XA->xlist; X.xcount = XA->xcount; X.ext = XA->ext; }
	return X;
}

#if defined(__HIPCC__)
__device__ __forceinline__ int rg_cal_max_gap(const RegParams &P, int qlen)   // memchain.c:576-582
{
	int l_del = (int)((double)(qlen * P.a - P.o_del) / P.e_del + 1.);
	int l_ins = (int)((double)(qlen * P.a - P.o_ins) / P.e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < P.w << 1 ? l : P.w << 1;
}
// cal_max_gap for every length a read of this kernel can ask about, tabulated once per workgroup (two double divisions each)
__device__ __forceinline__ int rg_gap(const int *tab, const RegParams &P, int qlen) { return (qlen >= 0 && qlen <= P.gap_cap) ? tab[qlen] : rg_cal_max_gap(P, qlen); }   // (gap_cap < 0: the kernel keeps no table)
#endif
