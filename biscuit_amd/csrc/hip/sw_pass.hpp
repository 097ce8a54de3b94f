// sw_pass.hpp -- one pass of ksw_u8 / ksw_i16 (lib/aln/ksw.c:111-334) by one wavefront: see k_sw.hip for how the striped
// SSE2 kernel's result is reproduced.  Shared by k_sw (K5, mate rescue) and the seed filter of long reads in k_regions.hip (C3).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

struct SwPass { int score, te, qe, score2, te2; };

template <int NC>
__device__ __forceinline__ SwPass sw_pass(const DevIndex &ix, const int8_t *mat, int is_u8, int qlen, const int (&qv)[NC],
                                          int tlen, long long tpos, int tdir, int te_rev,
                                          int o_del, int e_del, int o_ins, int e_ins, int xtra,
                                          unsigned long long *b, int lane)
{
	const int p = is_u8 ? 16 : 8, slen = (qlen + p - 1) / p, Q = slen * p, nch = (Q + 63) >> 6;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int minsc = (xtra & BSX_KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;
	const int endsc = (xtra & BSX_KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	int shift = 127, mx = 0;
	for (int a = 0; a < 25; ++a) { shift = mat[a] < shift ? mat[a] : shift; mx = mat[a] > mx ? mat[a] : mx; }
	shift = (int)(uint8_t)(256 - (int)(uint8_t)shift);   // ksw.c:84-88
	int Hp[NC], Hc[NC], E[NC], Hm[NC];
	// per column, fixed for the job: the scores of its query base against target bases 0..3 (a byte each; a padding column scores 0),
	// whether it opens a stripe, and its stripe number times seg_step.  The stripe-restricted F below is a prefix maximum that must not
	// look past the start of the column's stripe: with that added on top of every term (terms stay below seg_step) the
	// plain prefix maximum of the row can only be attained inside the column's own stripe, so both F's are DPP scans.
	uint32_t sqp[NC]; int segoff[NC]; bool headc[NC];
	const int seg_step = 1 << (32 - __builtin_clz((unsigned)(32768 + Q * e_ins)));   // above every term of the scans (h <= 32767, plus j * e_ins); 16 stripes stay far below 2^29
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		Hp[c] = 0; Hc[c] = 0; E[c] = 0; Hm[c] = 0;
		const int j = (c << 6) + lane, q = qv[c];
		sqp[c] = q > 4 ? 0u : ((uint32_t)(uint8_t)mat[q] | (uint32_t)(uint8_t)mat[5 + q] << 8 | (uint32_t)(uint8_t)mat[10 + q] << 16 | (uint32_t)(uint8_t)mat[15 + q] << 24);
		const int sg = j / slen;
		segoff[c] = sg * seg_step; headc[c] = j < Q && j - sg * slen == 0;
	}
	int gmax = 0, te = -1, n_b = 0;
	unsigned long long b_last = 0;
	int tb_reg = 4;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) {
			const int ii = i + lane;
			const long long src = (ii <= te_rev) ? (long long)(te_rev - ii) : (long long)ii;  // reversed prefix in pass 2
			tb_reg = ii < tlen ? dev_ref_base(ix.pac, ix.l_pac, tpos + src * tdir) : 4;
		}
		const int tsh = (wave_bcast(tb_reg, i & 63) & 3) << 3;   // the reference never holds an ambiguous base (bntseq.c:558-559)
		int rowmax = 0, pm_full = NEG_BIG, carry_seg = NEG_BIG;
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			if (c < nch) {
				const int j = (c << 6) + lane;
				const bool act = j < Q;
				int d = wave_prev(Hp[c], 0);
				if (lane == 0) d = 0;
				if (c > 0) { const int pv = __builtin_amdgcn_readlane(Hp[c > 0 ? c - 1 : 0], 63); if (lane == 0) d = pv; }
				const int s = (int)(int8_t)(sqp[c] >> tsh);
				int h;
				if (is_u8) { h = d + s + shift; h = h > 255 ? 255 : h; h -= shift; h = h < 0 ? 0 : h; }
				else { h = d + s; h = h > 32767 ? 32767 : h; }
				h = h > E[c] ? h : E[c];
				int tins = h - oe_ins; tins = tins > 0 ? tins : 0;
				const int g = act ? tins + j * e_ins : NEG_BIG;
				// full-row F
				const int incl = wave_scan_max_incl(g);
				int excl = wave_prev(incl, NEG_BIG);
				excl = excl > pm_full ? excl : pm_full;
				{ const int tot = __builtin_amdgcn_readlane(incl, 63); pm_full = pm_full > tot ? pm_full : tot; }
				int ff = j == 0 ? 0 : excl - (j - 1) * e_ins;
				ff = ff > 0 ? ff : 0;
				// stripe-restricted F (restarts at every multiple of slen)
				int sincl = wave_scan_max_incl(act ? g + segoff[c] : NEG_BIG);
				sincl = sincl > carry_seg ? sincl : carry_seg;
				const int sexcl = wave_prev(sincl, carry_seg);
				carry_seg = __builtin_amdgcn_readlane(sincl, 63);
				int fs = headc[c] ? 0 : sexcl - segoff[c] - (j - 1) * e_ins;
				fs = fs > 0 ? fs : 0;
				const int hpre = h > fs ? h : fs;
				const int hh = h > ff ? h : ff;
				int e = E[c] - e_del; e = e > 0 ? e : 0;
				int tdel = hpre - oe_del; tdel = tdel > 0 ? tdel : 0;
				e = e > tdel ? e : tdel;
				E[c] = act ? e : 0;
				Hc[c] = act ? hh : 0;
				rowmax = rowmax > Hc[c] ? rowmax : Hc[c];
			}
		}
		const int imax = wave_max_i32(rowmax);
		if (imax >= minsc) { // b[]: best (score,row) of each run of consecutive rows (ksw.c:192-200)
			if (n_b == 0 || (int)(uint32_t)b_last + 1 != i) { b_last = (unsigned long long)imax << 32 | (uint32_t)i; ++n_b; }
			else if ((int)(b_last >> 32) < imax) b_last = (unsigned long long)imax << 32 | (uint32_t)i;
			if (lane == 0) b[n_b - 1] = b_last;
		}
#pragma unroll
		for (int c = 0; c < NC; ++c) Hp[c] = Hc[c];
		if (imax > gmax) {
			gmax = imax; te = i;
#pragma unroll
			for (int c = 0; c < NC; ++c) Hm[c] = Hc[c];
			if ((is_u8 && gmax + shift >= 255) || gmax >= endsc) break;
		}
	}
	SwPass r;
	r.score = is_u8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
	r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
	if (!is_u8 || r.score != 255) {
		int lm = -1, lj = 0x7fffffff;
#pragma unroll
		for (int c = 0; c < NC; ++c) if (c < nch) { const int j = (c << 6) + lane; if (j < Q && Hm[c] > lm) { lm = Hm[c]; lj = j; } }
		const int m = wave_max_i32(lm);
		r.qe = wave_min_i32(lm == m ? lj : 0x7fffffff);   // smallest query index holding the maximum (ksw.c:212-216)
		if (n_b > 0) {
			WAVE_SYNC();
			const int rad = (r.score + mx - 1) / mx, low = te - rad, high = te + rad;
			int bs = -1, bi = 0x7fffffff;
			for (int k = lane; k < n_b; k += 64) {
				const unsigned long long v = b[k];
				const int e = (int)(uint32_t)v, sc = (int)(v >> 32);
				if ((e < low || e > high) && sc > bs) { bs = sc; bi = k; }
			}
			const int ms = wave_max_i32(bs);
			if (ms >= 0) {
				const int mi = wave_min_i32(bs == ms ? bi : 0x7fffffff);  // first entry with that score (ksw.c:222-225)
				r.score2 = ms; r.te2 = (int)(uint32_t)b[mi];
			}
			WAVE_SYNC();
		}
	}
	return r;
}

