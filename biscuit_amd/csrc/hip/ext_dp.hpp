// ext_dp.hpp -- the wavefront-wide ksw_extend2 (lib/aln/ksw.c:380-479) as a device function shared by
// k_extend (one job per wave) and k_regions (the chain->region state machine runs it inline).
// H/E/qb point to this wave's LDS: H,E >= qlen+2 int32 each, qb >= qlen bytes.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"

template <int NC>
__device__ __forceinline__ bsx_ext_res_t ext_dp(const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t &J,
                                                int32_t *H, int32_t *E, uint8_t *qb, int lane)
{
	const int qlen = J.qlen, tlen = J.tlen, h0 = J.h0;
	const int8_t *mat = J.parent ? sc.ctmat : sc.gamat;
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	// query into LDS; first row (ksw.c:395-397): H[0]=h0, H[j]=max(h0-oe_ins-(j-1)e_ins,0)
	for (int j = lane; j < qlen; j += 64) qb[j] = reads[(long long)J.qoff + (long long)j * J.qdir];
	for (int j = lane; j <= qlen; j += 64) {
		int v = j == 0 ? h0 : h0 - oe_ins - (j - 1) * e_ins;
		H[j] = v > 0 ? v : 0; E[j] = 0;
	}
	// band clamp (ksw.c:399-407)
	int mx = 0;
	for (int k = 0; k < 25; ++k) mx = mx > mat[k] ? mx : mat[k];
	int w = J.w;
	{
		int max_ins = (int)((double)(qlen * mx + J.end_bonus - o_ins) / e_ins + 1.);
		max_ins = max_ins > 1 ? max_ins : 1;
		w = w < max_ins ? w : max_ins;
		int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
		max_del = max_del > 1 ? max_del : 1;
		w = w < max_del ? w : max_del;
	}
	int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int tb_reg = 4;
	WAVE_SYNC();
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) tb_reg = (i + lane < tlen) ? dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)(i + lane) * J.tdir) : 4;
		const int t = wave_bcast(tb_reg, i & 63);
		const int8_t s0 = mat[t * 5], s1 = mat[t * 5 + 1], s2 = mat[t * 5 + 2], s3 = mat[t * 5 + 3], s4 = mat[t * 5 + 4];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1_init = 0;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		int m = 0, mj = -1, h1_last = h1_init;
		if (beg < end) {
			const int nch = (end - beg + 63) >> 6;
			int M[NC], Ev[NC];
			// pass A: read the whole row's inputs
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				M[c] = 0; Ev[c] = 0;
				if (c < nch) {
					const int j = beg + (c << 6) + lane;
					if (j < end) {
						const int hp = H[j], q = qb[j];
						const int s = q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : q == 3 ? s3 : s4;
						M[c] = hp ? hp + s : 0;
						Ev[c] = E[j];
					}
				}
			}
			WAVE_SYNC();
			// pass B: F by prefix scan, H, E', row maximum
			int carry = NEG_BIG, lm = -1, lj = -1;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				if (c < nch) {
					const int j = beg + (c << 6) + lane;
					const bool act = j < end;
					int tins = M[c] - oe_ins; tins = tins > 0 ? tins : 0;
					const int g = act ? tins + j * e_ins : NEG_BIG;
					const int incl = wave_scan_max_incl(g);
					int excl = wave_prev(incl, NEG_BIG);
					excl = excl > carry ? excl : carry;            // prefix max over all earlier columns
					{ const int tot = __builtin_amdgcn_readlane(incl, 63); carry = carry > tot ? carry : tot; }
					int f = j == beg ? 0 : excl - (j - 1) * e_ins;
					if (f < 0) f = 0;
					if (act) {
						int h = M[c] > Ev[c] ? M[c] : Ev[c];
						h = h > f ? h : f;
						int tdel = M[c] - oe_del; tdel = tdel > 0 ? tdel : 0;
						int e = Ev[c] - e_del; e = e > tdel ? e : tdel;
						E[j] = e;
						H[j + 1] = h;
						if (j == beg) H[beg] = h1_init;
						if (j == end - 1) E[end] = 0;
						if (h >= lm) { lm = h; lj = j; }
					}
				}
			}
			WAVE_SYNC();
			m = wave_max_i32(lm);
			mj = wave_max_i32(lm == m ? lj : -1);
			h1_last = H[end];
		} else { // empty row: only the boundary cell is written (ksw.c:449)
			if (lane == 0) { H[end] = h1_init; E[end] = 0; }
			WAVE_SYNC();
		}
		const int jfin = beg < end ? end : beg;
		if (jfin == qlen) { max_ie = gscore > h1_last ? max_ie : i; gscore = gscore > h1_last ? gscore : h1_last; }
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			int off = mj - i; off = off < 0 ? -off : off;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		// shrink the band to the non-zero cells (ksw.c:466-469), reading back what was written:
		// beg' = first j in [beg,end) with (H,E) != 0 (else end); end' = last such j in [beg',end] + 2
		{
			const int nchk = (end - beg + 1 + 63) >> 6;   // cells beg..end inclusive
			int nb = end, last;
			for (int c = 0; c < nchk; ++c) {
				const int j = beg + (c << 6) + lane;
				const bool nz = j < end && (H[j] != 0 || E[j] != 0);
				const unsigned long long b = __ballot(nz);
				if (b) { nb = beg + (c << 6) + __builtin_ctzll(b); break; }
			}
			last = nb - 1;
			for (int c = nchk - 1; c >= 0; --c) {
				const int j = beg + (c << 6) + lane;
				const bool nz = j <= end && j >= nb && (H[j] != 0 || E[j] != 0);
				const unsigned long long b = __ballot(nz);
				if (b) { last = beg + (c << 6) + 63 - __builtin_clzll(b); break; }
			}
			beg = nb;
			end = last + 2 < qlen ? last + 2 : qlen;
		}
		WAVE_SYNC();
	}
	bsx_ext_res_t r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	WAVE_SYNC();
	return r;
}

// The same recurrence with the two DP rows held in registers: lane l owns entries l, l+64, ... of the reference's
// eh[] array (H = the value ksw.c keeps in eh[j].h, i.e. the diagonal predecessor; E = eh[j].e), so a row costs no
// LDS traffic and no barriers: the right-shift of H by one column is a lane shift, F is the same max-plus prefix
// scan, and entries outside the band simply keep their old contents, exactly like the array they mirror.
// Needs qlen + 1 <= 64 * NC entries.
template <int NC>
__device__ __forceinline__ bsx_ext_res_t ext_dp_reg(const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t &J, int lane,
                                                    const uint8_t *win = nullptr, long long win_beg = 0, const uint8_t *qlds = nullptr, uint32_t qlds_off = 0)
{   // win: the reference bases [win_beg, ...) already in LDS, one byte each, covering every row of this job (else HBM)
    // qlds: the read already in LDS, qlds[k] = reads[qlds_off + k] (else the chunk's read buffer in HBM)
	const int qlen = J.qlen, tlen = J.tlen, h0 = J.h0;
	// the lane's column of this strand's 5x5 matrix: five byte loads per register slot from the kernel's argument block, once per
	// job (as 25 scalars selected per lane it was a hundred scalar instructions and 25 scalar registers in kernels that spill them)
	const int8_t *mat = J.parent ? sc.ctmat : sc.gamat;
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	int Hr[NC], Er[NC], sq4[NC];
	uint32_t sqp[NC];   // scores of this lane's query base of slot c against target bases 0..3, a byte each (the matrix is int8); sq4: against base 4
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const int a = (c << 6) + lane;
		const int q = a < qlen ? (qlds ? (int)qlds[(int)(J.qoff - qlds_off) + a * J.qdir] : (int)reads[(long long)J.qoff + (long long)a * J.qdir]) : 4;
		sqp[c] = (uint32_t)(uint8_t)mat[q] | (uint32_t)(uint8_t)mat[5 + q] << 8 | (uint32_t)(uint8_t)mat[10 + q] << 16 | (uint32_t)(uint8_t)mat[15 + q] << 24;   // q <= 4
		sq4[c] = mat[20 + q];
		const int v = a == 0 ? h0 : h0 - oe_ins - (a - 1) * e_ins;   // first row (ksw.c:395-397)
		Hr[c] = (a <= qlen && v > 0) ? v : 0;
		Er[c] = 0;
	}
	const int mx = J.parent ? sc.mx_ct : sc.mx_ga;
	int w = J.w;
	{ // band clamp (ksw.c:399-407)
		int max_ins = (int)((double)(qlen * mx + J.end_bonus - o_ins) / e_ins + 1.);
		max_ins = max_ins > 1 ? max_ins : 1;
		w = w < max_ins ? w : max_ins;
		int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
		max_del = max_del > 1 ? max_del : 1;
		w = w < max_del ? w : max_del;
	}
	int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int tb_reg = 4;
	const bool packed_max = qlen <= 512 && (long long)h0 + (long long)qlen * mx < (1 << 21);   // every H of this job fits 22 bits and every column 9
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) {
			const long long tp = J.tpos + (long long)(i + lane) * J.tdir;
			tb_reg = (i + lane < tlen) ? (win ? (int)win[tp - win_beg] : dev_ref_base(ix.pac, ix.l_pac, tp)) : 4;
		}
		const int t = wave_bcast(tb_reg, i & 63);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1_init = 0;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		int m = 0, mj = -1, h1_last = h1_init;
		if (beg < end) {
			const int c0 = NC == 1 ? 0 : beg >> 6, c1 = NC == 1 ? 0 : (end - 1) >> 6;   // one slot: every test on c folds away
			int hn[NC];
			int carry = NEG_BIG, lm = -1, lj = -1;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				hn[c] = 0;
				if (c >= c0 && c <= c1) {
					const int a = (c << 6) + lane;
					const bool act = a >= beg && a < end;
					const int s = t < 4 ? (int)(int8_t)(sqp[c] >> ((t & 3) << 3)) : sq4[c];   // one bit-field extract by a scalar shift
					const int M = (act && Hr[c]) ? Hr[c] + s : 0;
					int tins = M - oe_ins; tins = tins > 0 ? tins : 0;
					const int g = act ? tins + a * e_ins : NEG_BIG;
					const int incl = wave_scan_max_incl(g);
					int excl = wave_prev(incl, NEG_BIG);
					excl = excl > carry ? excl : carry;
					{ const int tot = __builtin_amdgcn_readlane(incl, 63); carry = carry > tot ? carry : tot; }
					int f = a == beg ? 0 : excl - (a - 1) * e_ins;
					if (f < 0) f = 0;
					if (act) {
						int h = M > Er[c] ? M : Er[c];
						h = h > f ? h : f;
						int tdel = M - oe_del; tdel = tdel > 0 ? tdel : 0;
						int e = Er[c] - e_del; e = e > tdel ? e : tdel;
						Er[c] = e;
						hn[c] = h;
						if (h >= lm) { lm = h; lj = a; }
					} else if (a == end) Er[c] = 0;
				}
			}
			// eh[end].e = 0 when `end` opens a chunk the loop above did not visit
			if (NC > 1 && (end & 63) == 0 && (end >> 6) < NC && (end >> 6) > c1) {
#pragma unroll
				for (int c = 0; c < NC; ++c) if (c == (end >> 6) && lane == 0) Er[c] = 0;
			}
			// H: entry a takes h(i, a-1) for a-1 in the band, entry beg takes the first-column value
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (c >= c0 && c <= c1 + 1) {
					const int a = (c << 6) + lane;
					int up = wave_prev(hn[c], 0);
					const int edge = c > 0 ? __builtin_amdgcn_readlane(hn[c > 0 ? c - 1 : 0], 63) : 0;
					if (lane == 0) up = edge;
					if (a == beg) Hr[c] = h1_init;
					else if (a - 1 >= beg && a - 1 < end) Hr[c] = up;
				}
			}
			if (packed_max) { // row maximum and the last column that attains it in one reduction: (h << 9 | column), columns < 512
				const int key = wave_max_i32(lm < 0 ? -1 : (lm << 9 | lj));
				m = key >> 9; mj = key < 0 ? -1 : (key & 511);
			} else {
				m = wave_max_i32(lm);
				mj = wave_max_i32(lm == m ? lj : -1);
			}
			if (end == qlen) { // h(i, end-1), only read for the to-the-end score
				int v = 0;
#pragma unroll
				for (int c = 0; c < NC; ++c) if (c == c1) v = hn[c];
				h1_last = wave_bcast(v, (end - 1) & 63);
			}
		} else { // empty row: only the boundary cell is written (ksw.c:449)
#pragma unroll
			for (int c = 0; c < NC; ++c) if ((c << 6) + lane == end) { Hr[c] = h1_init; Er[c] = 0; }
		}
		const int jfin = beg < end ? end : beg;
		if (jfin == qlen) { max_ie = gscore > h1_last ? max_ie : i; gscore = gscore > h1_last ? gscore : h1_last; }
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			int off = mj - i; off = off < 0 ? -off : off;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		// shrink the band to the non-zero cells (ksw.c:466-469)
		{
			int nb = end, last;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				if (NC == 1 || (nb == end && c >= (beg >> 6) && c <= ((end - 1) >> 6))) {
					const int a = (c << 6) + lane;
					const unsigned long long b = __ballot(a >= beg && a < end && (Hr[c] != 0 || Er[c] != 0));
					if (b) nb = (c << 6) + __builtin_ctzll(b);
				}
			}
			last = nb - 1;
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (NC == 1 || (last == nb - 1 && c <= (end >> 6) && c >= (nb >> 6))) {
					const int a = (c << 6) + lane;
					const unsigned long long b = __ballot(a <= end && a >= nb && (Hr[c] != 0 || Er[c] != 0));
					if (b) last = (c << 6) + 63 - __builtin_clzll(b);
				}
			}
			beg = nb;
			end = last + 2 < qlen ? last + 2 : qlen;
		}
	}
	bsx_ext_res_t r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// The same rows in registers for queries of ANY length (round 6): a WINDOW of NW slots of 64 columns that follows the band.  Lane l of window slot
// c owns entry ((cb + c) << 6) + l of the reference's eh[] array; the band of row i is [i - w, i + w + 1) cut to the non-zero cells, at most 2 w + 1
// columns, so with cb = the slot of its first column it lies inside NW = ceil((2 w + 1) / 64) + 1 slots (five for w <= 127), column `end` included.
// When the band's first column leaves slot cb the window moves up a slot: registers renamed, and the slot that comes in takes what eh[] holds
// where no row has been yet -- the first row's values (ksw.c:395-397): every column the band has visited or written (the last one is eh[end])
// is at or below the previous `end`, which the window held.  A kilobase read's extension (500 rows, a band of 201 columns) was the LDS form above:
// three barriers and two dozen LDS round trips a row at one wave per SIMD, 4 900 cycles a row, 98 % of the long reads' chains -> regions launch.
// Needs 2 * w + 1 <= 64 * (NW - 1) after the band clamp (the caller tests J.w, which only shrinks) and a score below 2^21.
template <int NW>
__device__ __forceinline__ bsx_ext_res_t ext_dp_win(const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t &J, int lane,
                                                    const uint8_t *win = nullptr, long long win_beg = 0, const uint8_t *qlds = nullptr, uint32_t qlds_off = 0)
{
	const int qlen = J.qlen, tlen = J.tlen, h0 = J.h0;
	const int8_t *mat = J.parent ? sc.ctmat : sc.gamat;
	const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
	int Hr[NW], Er[NW], sq4[NW];
	uint32_t sqp[NW];
	int cb = 0;   // the window's first slot
#define EXT_WIN_LOAD(c_, slot_) do { const int a_ = ((slot_) << 6) + lane; \
		const int q_ = a_ < qlen ? (qlds ? (int)qlds[(int)(J.qoff - qlds_off) + a_ * J.qdir] : (int)reads[(long long)J.qoff + (long long)a_ * J.qdir]) : 4; \
		sqp[c_] = (uint32_t)(uint8_t)mat[q_] | (uint32_t)(uint8_t)mat[5 + q_] << 8 | (uint32_t)(uint8_t)mat[10 + q_] << 16 | (uint32_t)(uint8_t)mat[15 + q_] << 24; \
		sq4[c_] = mat[20 + q_]; \
		const int v_ = a_ == 0 ? h0 : h0 - oe_ins - (a_ - 1) * e_ins; \
		Hr[c_] = (a_ <= qlen && v_ > 0) ? v_ : 0; Er[c_] = 0; } while (0)
#pragma unroll
	for (int c = 0; c < NW; ++c) EXT_WIN_LOAD(c, c);
	const int mx = J.parent ? sc.mx_ct : sc.mx_ga;
	int w = J.w;
	{ // band clamp (ksw.c:399-407)
		int max_ins = (int)((double)(qlen * mx + J.end_bonus - o_ins) / e_ins + 1.);
		max_ins = max_ins > 1 ? max_ins : 1;
		w = w < max_ins ? w : max_ins;
		int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
		max_del = max_del > 1 ? max_del : 1;
		w = w < max_del ? w : max_del;
	}
	int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int tb_reg = 4;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) {
			const long long tp = J.tpos + (long long)(i + lane) * J.tdir;
			tb_reg = (i + lane < tlen) ? (win ? (int)win[tp - win_beg] : dev_ref_base(ix.pac, ix.l_pac, tp)) : 4;
		}
		const int t = wave_bcast(tb_reg, i & 63);
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		while ((beg >> 6) > cb) { // the window follows the band, a slot at a time
#pragma unroll
			for (int c = 0; c + 1 < NW; ++c) { Hr[c] = Hr[c + 1]; Er[c] = Er[c + 1]; sqp[c] = sqp[c + 1]; sq4[c] = sq4[c + 1]; }
			++cb;
			EXT_WIN_LOAD(NW - 1, cb + NW - 1);
		}
		int h1_init = 0;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		int m = 0, mj = -1, h1_last = h1_init;
		if (beg < end) {
			const int c0 = (beg >> 6) - cb, c1 = ((end - 1) >> 6) - cb;   // window slots of the band's first and last column
			int hn[NW];
			int carry = NEG_BIG, lm = -1, lj = -1;
#pragma unroll
			for (int c = 0; c < NW; ++c) {
				hn[c] = 0;
				if (c >= c0 && c <= c1) {
					const int a = ((cb + c) << 6) + lane;
					const bool act = a >= beg && a < end;
					const int s = t < 4 ? (int)(int8_t)(sqp[c] >> ((t & 3) << 3)) : sq4[c];
					const int M = (act && Hr[c]) ? Hr[c] + s : 0;
					int tins = M - oe_ins; tins = tins > 0 ? tins : 0;
					const int g = act ? tins + a * e_ins : NEG_BIG;
					const int incl = wave_scan_max_incl(g);
					int excl = wave_prev(incl, NEG_BIG);
					excl = excl > carry ? excl : carry;
					{ const int tot = __builtin_amdgcn_readlane(incl, 63); carry = carry > tot ? carry : tot; }
					int f = a == beg ? 0 : excl - (a - 1) * e_ins;
					if (f < 0) f = 0;
					if (act) {
						int h = M > Er[c] ? M : Er[c];
						h = h > f ? h : f;
						int tdel = M - oe_del; tdel = tdel > 0 ? tdel : 0;
						int e = Er[c] - e_del; e = e > tdel ? e : tdel;
						Er[c] = e;
						hn[c] = h;
						if (h >= lm) { lm = h; lj = a; }
					} else if (a == end) Er[c] = 0;
				}
			}
			// eh[end].e = 0 when `end` opens a slot the loop above did not visit
			if ((end & 63) == 0 && (end >> 6) - cb > c1 && (end >> 6) - cb < NW) {
#pragma unroll
				for (int c = 0; c < NW; ++c) if (c == (end >> 6) - cb && lane == 0) Er[c] = 0;
			}
			// H: entry a takes h(i, a-1) for a-1 in the band, entry beg takes the first-column value
#pragma unroll
			for (int c = NW - 1; c >= 0; --c) {
				if (c >= c0 && c <= c1 + 1) {
					const int a = ((cb + c) << 6) + lane;
					int up = wave_prev(hn[c], 0);
					const int edge = c > 0 ? __builtin_amdgcn_readlane(hn[c > 0 ? c - 1 : 0], 63) : 0;
					if (lane == 0) up = edge;
					if (a == beg) Hr[c] = h1_init;
					else if (a - 1 >= beg && a - 1 < end) Hr[c] = up;
				}
			}
			{ // row maximum and the last column that attains it in one reduction: (h << 9 | column inside the window), 64 NW <= 512 columns
				const int key = wave_max_i32(lm < 0 ? -1 : (lm << 9 | (lj - (cb << 6))));
				m = key >> 9; mj = key < 0 ? -1 : (key & 511) + (cb << 6);
			}
			if (end == qlen) { // h(i, end-1), only read for the to-the-end score
				int v = 0;
#pragma unroll
				for (int c = 0; c < NW; ++c) if (c == c1) v = hn[c];
				h1_last = wave_bcast(v, (end - 1) & 63);
			}
		} else { // empty row: only the boundary cell is written (ksw.c:449)
#pragma unroll
			for (int c = 0; c < NW; ++c) if (((cb + c) << 6) + lane == end) { Hr[c] = h1_init; Er[c] = 0; }
		}
		const int jfin = beg < end ? end : beg;
		if (jfin == qlen) { max_ie = gscore > h1_last ? max_ie : i; gscore = gscore > h1_last ? gscore : h1_last; }
		if (m == 0) break;
		if (m > max) {
			max = m; max_i = i; max_j = mj;
			int off = mj - i; off = off < 0 ? -off : off;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
			else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
		}
		// shrink the band to the non-zero cells (ksw.c:466-469)
		{
			int nb = end, last;
#pragma unroll
			for (int c = 0; c < NW; ++c) {
				if (nb == end && cb + c >= (beg >> 6) && cb + c <= ((end - 1) >> 6)) {
					const int a = ((cb + c) << 6) + lane;
					const unsigned long long b = __ballot(a >= beg && a < end && (Hr[c] != 0 || Er[c] != 0));
					if (b) nb = ((cb + c) << 6) + __builtin_ctzll(b);
				}
			}
			last = nb - 1;
#pragma unroll
			for (int c = NW - 1; c >= 0; --c) {
				if (last == nb - 1 && cb + c <= (end >> 6) && cb + c >= (nb >> 6)) {
					const int a = ((cb + c) << 6) + lane;
					const unsigned long long b = __ballot(a <= end && a >= nb && (Hr[c] != 0 || Er[c] != 0));
					if (b) last = ((cb + c) << 6) + 63 - __builtin_clzll(b);
				}
			}
			beg = nb;
			end = last + 2 < qlen ? last + 2 : qlen;
		}
	}
#undef EXT_WIN_LOAD
	bsx_ext_res_t r;
	r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}
