// dev_common.hpp -- device-side views of the resident index and shared helpers (gfx950 only).
// BSX_HD functions are plain C++ so the per-lane logic can also be compiled by g++ for the
// CPU-side kernel-logic tests (tests/test_kernel_logic_host.py); the kernels themselves are HIP.
#pragma once
#include <stdint.h>
#include "bsx.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BSX_HD __host__ __device__ __forceinline__
#define BSX_D  __device__ __forceinline__
#else
#define BSX_HD inline
#define BSX_D  inline
struct uint4 { uint32_t x, y, z, w; };
#endif

// FM index of one converted text, resident in HBM.  Block b (128 symbols) = words [16b,16b+16):
// 4 x u64 cumulative counts then 8 x u32 of 2-bit symbols, first symbol in the top bits
// (bwt_t, lib/aln/bwt.h:54-71,93-101).
struct DevFmi {
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	const uint32_t *bwt;
	const uint64_t *sa;
	uint32_t sa_mask;      // sa_intv - 1
	uint32_t sa_shift;     // log2(sa_intv)
};

// bi-intervals of every string of up to K letters of a converted read's three-letter alphabet (seed_tab.hpp): 16-byte entries
// (x0, x1, x2 low words, high bits), level L at ((3^L - 3) >> 1), a string's letters as base-3 digits, first letter most significant
struct DevSeedTab { const uint4 *t[2]; int32_t K; int32_t pad_; };   // [1] parent, [0] daughter; K = 0: none

struct DevIndex {
	DevFmi fmi[2];         // [1] parent, [0] daughter
	const uint8_t *pac;
	int64_t l_pac;
	const int64_t *ctg_off;    // n_seqs + 1 contig offsets on the forward strand (bntann1_t.offset; [n_seqs] = l_pac)
	const uint8_t *ctg_alt;    // n_seqs: is_alt
	int32_t n_seqs;
	DevSeedTab tab;
};

// the options the region kernel reads (mem_opt_t fields of the same name)
struct RegParams {
	int32_t a, w, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3;
	int32_t min_seed_len, min_chain_weight, max_chain_gap, max_occ, bsstrand;
	uint32_t max_chain_extend;
	float mask_level, drop_ratio;
	int32_t prof;          // count wave cycles per stage (tracing only: the counters are contended atomics)
	int32_t walk_on;       // the HBM tiers walk an over-represented interval past max_occ themselves (0: such strand searches are left to the caller)
	int32_t ext_win;       // chains -> regions of long reads: an extension's rows in a register window that follows the band (ext_dp_win) instead of in LDS
	int32_t gap_cap;       // cal_max_gap is tabulated up to this query length (set by each kernel to the size of its table)
	int32_t flt_len;       // flt_tab has entries for read lengths 0..flt_len
	const int32_t *flt_tab; // per read length: min_HSP_score of mem_flt_chained_seeds (memchain.c:544-548) when the seed-SW filter runs for
	                       // reads of that length, INT32_MIN when it does not (made by the host: the rule goes through log())
};

struct DevScoring {        // set by bsx_device_set_opt
	int8_t ctmat[25];
	int8_t gamat[25];
	int32_t o_del, e_del, o_ins, e_ins, zdrop, a;
	int32_t mx_ct, mx_ga;      // the largest entry of each matrix (ksw_extend2's band clamp, ksw.c:399-407)
};

// reference base at forward-reverse coordinate p (bns_get_seq, lib/aln/bntseq.c:402-422)
BSX_HD int dev_ref_base(const uint8_t *pac, int64_t l_pac, int64_t p)
{
	if (p >= l_pac) { p = (l_pac << 1) - 1 - p; return 3 - ((pac[p >> 2] >> ((~p & 3) << 1)) & 3); }
	return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

// bns_fetch_seq (memchain.c:889) of [beg, beg + span) into LDS, one byte per base: the window lies on one strand (the caller
// clamps it at l_pac, memchain.c:606-610), i.e. it is a run of consecutive 2-bit codes of pac read forwards, or backwards and
// complemented.  A lane takes an aligned dword of pac (16 bases): one load instruction for the whole window instead of one
// dependent round trip to HBM per 64 bases.
BSX_HD void dev_fetch_window(uint8_t *win, const uint8_t *pac, int64_t l_pac, int64_t beg, int span, int lane)
{
	if (beg < l_pac && beg + span > l_pac) { // not reached by the callers; the per-base form handles any window
		for (int i = lane; i < span; i += 64) win[i] = (uint8_t)dev_ref_base(pac, l_pac, beg + i);
		return;
	}
	const bool rev = beg >= l_pac;
	const int64_t f_lo = rev ? (l_pac << 1) - (beg + span) : beg;   // forward coordinates [f_lo, f_lo + span)
	const int64_t b0 = (f_lo >> 2) & ~(int64_t)3;                          // first byte of pac touched, rounded down to a dword (pac is padded)
	const int nw = (int)((((f_lo + span - 1) >> 2) - b0) >> 2) + 1;
	for (int w = lane; w < nw; w += 64) {
		const uint8_t *pw = pac + b0 + (int64_t)4 * w;
		const uint32_t v = *reinterpret_cast<const uint32_t*>(pw);
		const int64_t f0 = (b0 + (int64_t)4 * w) << 2;
#pragma unroll
		for (int k = 0; k < 16; ++k) { // base k of the dword: byte k >> 2, bits (~k & 3) << 1 (bntseq.h _get_pac)
			const int b = (int)(v >> (8 * (k >> 2) + ((~k & 3) << 1))) & 3;
			const int64_t i = rev ? (f_lo + span - 1) - (f0 + k) : (f0 + k) - f_lo;
			if (i >= 0 && i < span) win[i] = (uint8_t)(rev ? 3 - b : b);
		}
	}
}

BSX_HD int dev_popc(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __popc(x);
#else
	return __builtin_popcount(x);
#endif
}

// counts of A,C,G,T among symbols 0..upto (inclusive) of the 8 symbol words of one block,
// packed as 4 x 8 bits... kept as four ints to stay simple: max 128 each.
BSX_HD void dev_block_count(const uint32_t w[8], int upto, uint32_t cnt[4])
{
	uint32_t a = 0, c = 0, g = 0, t = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		int nv = upto + 1 - 16 * i;
		nv = nv < 0 ? 0 : (nv > 16 ? 16 : nv);
		// top nv symbols valid
		uint32_t valid = nv == 0 ? 0u : (0x55555555u & ~(uint32_t)((1ull << ((16 - nv) << 1)) - 1));
		uint32_t lo = w[i] & valid, hi = (w[i] >> 1) & valid;
		uint32_t nt = (uint32_t)dev_popc(hi & lo), ng = (uint32_t)dev_popc(hi & ~lo), nc = (uint32_t)dev_popc(lo & ~hi);
		t += nt; g += ng; c += nc; a += (uint32_t)nv - nt - ng - nc;
	}
	cnt[0] = a; cnt[1] = c; cnt[2] = g; cnt[3] = t;
}

struct DevIntv { uint64_t x0, x1, x2, info; };   // bwtintv_t

// Load one 64-byte block: counts (4 x u64) + symbol words.
BSX_HD void dev_load_block(const uint32_t *bwt, uint64_t kadj, uint64_t base[4], uint32_t w[8])
{
	const uint4 *p = reinterpret_cast<const uint4*>(bwt + ((kadj >> 7) << 4));
	uint4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
	base[0] = (uint64_t)v0.y << 32 | v0.x; base[1] = (uint64_t)v0.w << 32 | v0.z;
	base[2] = (uint64_t)v1.y << 32 | v1.x; base[3] = (uint64_t)v1.w << 32 | v1.z;
	w[0] = v2.x; w[1] = v2.y; w[2] = v2.z; w[3] = v2.w; w[4] = v3.x; w[5] = v3.y; w[6] = v3.z; w[7] = v3.w;
}

// bwt_2occ4 (lib/aln/bwt.c:204-236): ranks of all four symbols at k and l.
// returns 1 when the reference would take its one-block fast path (one 64-B touch), else 0 (two).
struct DevBlock { uint4 v0, v1, v2, v3; };
BSX_HD DevBlock dev_load_block4(const uint32_t *bwt, uint64_t kadj)
{
	const uint4 *p = reinterpret_cast<const uint4*>(bwt + ((kadj >> 7) << 4));
	DevBlock b; b.v0 = p[0]; b.v1 = p[1]; b.v2 = p[2]; b.v3 = p[3];
	return b;
}
// per-symbol counts among symbols 0..upto of the block: 4 x 8-bit packed (A | C<<8 | G<<16 | T<<24), max 128 each
BSX_HD uint32_t dev_count_word(uint32_t x, int nv)
{
	nv = nv < 0 ? 0 : (nv > 16 ? 16 : nv);
	const uint32_t valid = nv == 0 ? 0u : (0x55555555u & ~(uint32_t)((1ull << ((16 - nv) << 1)) - 1));
	const uint32_t lo = x & valid, hi = (x >> 1) & valid;
	const uint32_t nt = (uint32_t)dev_popc(hi & lo), ng = (uint32_t)dev_popc(hi & ~lo), nc = (uint32_t)dev_popc(lo & ~hi);
	return ((uint32_t)nv - nt - ng - nc) | nc << 8 | ng << 16 | nt << 24;
}
// T, G and C are counted word by word straight into three accumulators (popcount-and-add is one instruction); A is what is
// left of the upto+1 symbols.  nv = how many symbols of the word lie at or before `upto` (the first symbol sits in the top bits).
BSX_HD void dev_count_tgc(uint32_t x, int nv, uint32_t &nt, uint32_t &ng, uint32_t &nc)
{
	nv = nv < 0 ? 0 : (nv > 16 ? 16 : nv);
	// the low bit of each of the nv leading symbols: a 64-bit shift has no special case at either end (0 and 16 symbols)
	const uint32_t valid = (uint32_t)(0xffffffff00000000ull >> (nv << 1)) & 0x55555555u;
	const uint32_t lo = x & valid, hi = (x >> 1) & valid, both = hi & lo;
	nt += (uint32_t)dev_popc(both); ng += (uint32_t)dev_popc(hi ^ both); nc += (uint32_t)dev_popc(lo ^ both);
}
BSX_HD void dev_block_count4(const DevBlock &b, int upto, uint32_t &a, uint32_t &c, uint32_t &g, uint32_t &t)
{
	const int n = upto + 1;
	uint32_t nt = 0, ng = 0, nc = 0;
	dev_count_tgc(b.v2.x, n, nt, ng, nc);       dev_count_tgc(b.v2.y, n - 16, nt, ng, nc);
	dev_count_tgc(b.v2.z, n - 32, nt, ng, nc);  dev_count_tgc(b.v2.w, n - 48, nt, ng, nc);
	dev_count_tgc(b.v3.x, n - 64, nt, ng, nc);  dev_count_tgc(b.v3.y, n - 80, nt, ng, nc);
	dev_count_tgc(b.v3.z, n - 96, nt, ng, nc);  dev_count_tgc(b.v3.w, n - 112, nt, ng, nc);
	a = (uint32_t)n - nt - ng - nc; c = nc; g = ng; t = nt;
}

// ---- the device's own block layout.  In HBM the 128 symbols of a block are two BIT PLANES instead of the file's 2-bit fields: words 8-11
// hold the low bit of symbols 0-31, 32-63, 64-95, 96-127 (the first symbol in the top bit), words 12-15 the high bit (k_bwt_planes turns
// the file layout into this one, in place, right after the upload or the build).  A rank query then masks four words per plane with a
// prefix mask and takes twelve population counts -- 40 vector instructions where the 2-bit fields cost 104 (a shift, two masks and a
// combination per 16-symbol word), and the seeding kernel is bound by exactly those: a quarter of its instructions were the two block
// counts of an extension.  The cumulative counts in words 0-7 are the file's.
// NOT part of the device image: the file's final 8-word cumulative-count tail.  It sits right behind the last, partial block's symbol
// words, i.e. inside the 16 words that block takes in this layout, and k_bwt_planes shuffles it along with them.  Nothing on the device
// reads it (a rank query masks the planes to positions before its own, and the totals are in DevFmi::L2); a copy of the device's bwt back
// to the host is not a .bwt file body.
BSX_HD void dev_planes_count4(const DevBlock &b, int upto, uint32_t &a, uint32_t &c, uint32_t &g, uint32_t &t)
{
	const int n = upto + 1;
	uint32_t nt = 0, nh = 0, nl = 0;
#define BSX_PLANE_WORD(L_, H_, off_) do { int nv = n - (off_); nv = nv < 0 ? 0 : (nv > 32 ? 32 : nv); \
		const uint32_t m = (uint32_t)(0xffffffff00000000ull >> nv), lo = (L_) & m, hi = (H_) & m; \
		nl += (uint32_t)dev_popc(lo); nh += (uint32_t)dev_popc(hi); nt += (uint32_t)dev_popc(lo & hi); } while (0)
	BSX_PLANE_WORD(b.v2.x, b.v3.x, 0); BSX_PLANE_WORD(b.v2.y, b.v3.y, 32); BSX_PLANE_WORD(b.v2.z, b.v3.z, 64); BSX_PLANE_WORD(b.v2.w, b.v3.w, 96);
#undef BSX_PLANE_WORD
	t = nt; g = nh - nt; c = nl - nt; a = (uint32_t)n - nh - nl + nt;
}
BSX_HD int dev_planes_symbol(const DevBlock &b, int pos)   // the symbol at position pos (0..127) of the block
{
	const int p = pos >> 5, bit = 31 - (pos & 31);
	const uint32_t lo = p == 0 ? b.v2.x : p == 1 ? b.v2.y : p == 2 ? b.v2.z : b.v2.w, hi = p == 0 ? b.v3.x : p == 1 ? b.v3.y : p == 2 ? b.v3.z : b.v3.w;
	return (int)(((hi >> bit) & 1u) << 1 | ((lo >> bit) & 1u));
}

// bwt_2occ4 (lib/aln/bwt.c:204-236): ranks of all four symbols at k and l (scalars, no indexed arrays).
// returns 1 when the reference would take its one-block fast path (one 64-B touch), else 0 (two).
BSX_HD int dev_2occ4(const DevFmi &f, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	const uint64_t NEG1 = ~0ull;
	const uint64_t ka = k - (k >= f.primary), la = l - (l >= f.primary);
	const bool kv = k != NEG1, lv = l != NEG1;
	const bool same = kv && lv && (ka >> 7) == (la >> 7);
	// both gathers are issued before either is consumed (k == -1 reads block 0 and is discarded)
	const DevBlock B0 = dev_load_block4(f.bwt, kv ? ka : 0);
	DevBlock B1 = B0;
	if (!same) B1 = dev_load_block4(f.bwt, lv ? la : 0);
	uint32_t a, c, g, t;
	dev_block_count4(B0, (int)(ka & 127), a, c, g, t);
	ck[0] = kv ? ((uint64_t)B0.v0.y << 32 | B0.v0.x) + a : 0; ck[1] = kv ? ((uint64_t)B0.v0.w << 32 | B0.v0.z) + c : 0;
	ck[2] = kv ? ((uint64_t)B0.v1.y << 32 | B0.v1.x) + g : 0; ck[3] = kv ? ((uint64_t)B0.v1.w << 32 | B0.v1.z) + t : 0;
	dev_block_count4(B1, (int)(la & 127), a, c, g, t);
	cl[0] = lv ? ((uint64_t)B1.v0.y << 32 | B1.v0.x) + a : 0; cl[1] = lv ? ((uint64_t)B1.v0.w << 32 | B1.v0.z) + c : 0;
	cl[2] = lv ? ((uint64_t)B1.v1.y << 32 | B1.v1.x) + g : 0; cl[3] = lv ? ((uint64_t)B1.v1.w << 32 | B1.v1.z) + t : 0;
	return same ? 1 : 0;
}

// the same over the device's bit-plane blocks (the table builder of k_seedt.hip: a thread per extension)
BSX_HD void dev_2occ4_planes(const DevFmi &f, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
	const uint64_t NEG1 = ~0ull;
	const uint64_t ka = k - (k >= f.primary), la = l - (l >= f.primary);
	const bool kv = k != NEG1, lv = l != NEG1;
	const bool same = kv && lv && (ka >> 7) == (la >> 7);
	const DevBlock B0 = dev_load_block4(f.bwt, kv ? ka : 0);
	DevBlock B1 = B0;
	if (!same) B1 = dev_load_block4(f.bwt, lv ? la : 0);
	uint32_t a, c, g, t;
	dev_planes_count4(B0, (int)(ka & 127), a, c, g, t);
	ck[0] = kv ? ((uint64_t)B0.v0.y << 32 | B0.v0.x) + a : 0; ck[1] = kv ? ((uint64_t)B0.v0.w << 32 | B0.v0.z) + c : 0;
	ck[2] = kv ? ((uint64_t)B0.v1.y << 32 | B0.v1.x) + g : 0; ck[3] = kv ? ((uint64_t)B0.v1.w << 32 | B0.v1.z) + t : 0;
	dev_planes_count4(B1, (int)(la & 127), a, c, g, t);
	cl[0] = lv ? ((uint64_t)B1.v0.y << 32 | B1.v0.x) + a : 0; cl[1] = lv ? ((uint64_t)B1.v0.w << 32 | B1.v0.z) + c : 0;
	cl[2] = lv ? ((uint64_t)B1.v1.y << 32 | B1.v1.x) + g : 0; cl[3] = lv ? ((uint64_t)B1.v1.w << 32 | B1.v1.z) + t : 0;
}

// One of the two resident indices picked by value with selects: indexing the kernel-argument struct with a
// per-lane index would make the compiler copy it to scratch memory.
BSX_HD DevFmi dev_fmi_pick(const DevIndex &ix, int which)
{
	DevFmi f;
	f.primary = which ? ix.fmi[1].primary : ix.fmi[0].primary;
	f.L2[0] = which ? ix.fmi[1].L2[0] : ix.fmi[0].L2[0]; f.L2[1] = which ? ix.fmi[1].L2[1] : ix.fmi[0].L2[1];
	f.L2[2] = which ? ix.fmi[1].L2[2] : ix.fmi[0].L2[2]; f.L2[3] = which ? ix.fmi[1].L2[3] : ix.fmi[0].L2[3];
	f.L2[4] = which ? ix.fmi[1].L2[4] : ix.fmi[0].L2[4];
	f.seq_len = which ? ix.fmi[1].seq_len : ix.fmi[0].seq_len;
	f.bwt = which ? ix.fmi[1].bwt : ix.fmi[0].bwt; f.sa = which ? ix.fmi[1].sa : ix.fmi[0].sa;
	f.sa_mask = ix.fmi[0].sa_mask; f.sa_shift = ix.fmi[0].sa_shift;
	return f;
}
// field selects straight from the (uniform) kernel argument: nothing of the index is kept in per-lane registers
BSX_HD uint64_t dev_ix_L2(const DevIndex &ix, int which, int c)
{
	const uint64_t a = c == 0 ? ix.fmi[0].L2[0] : c == 1 ? ix.fmi[0].L2[1] : c == 2 ? ix.fmi[0].L2[2] : c == 3 ? ix.fmi[0].L2[3] : ix.fmi[0].L2[4];
	const uint64_t b = c == 0 ? ix.fmi[1].L2[0] : c == 1 ? ix.fmi[1].L2[1] : c == 2 ? ix.fmi[1].L2[2] : c == 3 ? ix.fmi[1].L2[3] : ix.fmi[1].L2[4];
	return which ? b : a;
}
BSX_HD uint64_t dev_ix_primary(const DevIndex &ix, int which) { return which ? ix.fmi[1].primary : ix.fmi[0].primary; }
BSX_HD const uint32_t *dev_ix_bwt(const DevIndex &ix, int which) { return which ? ix.fmi[1].bwt : ix.fmi[0].bwt; }

BSX_HD uint64_t dev_L2(const DevFmi &f, int c) { return c == 0 ? f.L2[0] : c == 1 ? f.L2[1] : c == 2 ? f.L2[2] : c == 3 ? f.L2[3] : f.L2[4]; }

// bwt_extend (lib/aln/bwt.c:278-293), returning only the child interval for symbol c.
BSX_HD DevIntv dev_extend(const DevFmi &f, const DevIntv &ik, int is_back, int c, uint32_t &n_slow, uint32_t &n_fast)
{
	uint64_t tk[4], tl[4];
	uint64_t xa = is_back ? ik.x0 : ik.x1;   // x[!is_back]
	uint64_t xb = is_back ? ik.x1 : ik.x0;   // x[is_back]
	if (dev_2occ4(f, xa - 1, xa - 1 + ik.x2, tk, tl)) ++n_fast; else ++n_slow;
	uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1], s0 = tl[0] - tk[0];
	uint64_t b3 = xb + ((xa <= f.primary && xa + ik.x2 - 1 >= f.primary) ? 1 : 0);
	uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const uint64_t tkc = c == 3 ? tk[3] : c == 2 ? tk[2] : c == 1 ? tk[1] : tk[0];
	uint64_t na = dev_L2(f, c) + 1 + tkc;
	uint64_t nb = c == 3 ? b3 : c == 2 ? b2 : c == 1 ? b1 : b0;
	uint64_t ns = c == 3 ? s3 : c == 2 ? s2 : c == 1 ? s1 : s0;
	DevIntv o;
	o.x0 = is_back ? na : nb; o.x1 = is_back ? nb : na; o.x2 = ns; o.info = 0;
	return o;
}
