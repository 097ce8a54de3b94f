// k_regions.hip -- the middle of the pipeline fused on the device: one wavefront per strand search takes
// the SA intervals of K1+K2 all the way to alignment regions:
//   K3  bwt_sa for every occurrence (lanes in parallel)                       lib/aln/bwt.c:87-97
//   C1  mem_chain: seeds clustered into chains in reference order             lib/aln/memchain.c:268-393
//   C2  mem_chain_flt (weights, klib introsort permutation, overlap filter)   lib/aln/memchain.c:406-488
//   C4  mem_chain2region(1): best-first seed extension with the containment   lib/aln/memchain.c:742-904
//       tests, left/right ksw_extend2 (wave-wide, ext_dp.hpp) and band retries
// Everything lives in this wave's LDS; nothing goes back to the host between seeding and regions.
//
// The kernel only takes the common case and says so per task (status != 0 => the host runs its own
// C1/C2/C4 for that task through the batch kernels): more than RG_SCAP occurrences / RG_CCAP chains /
// RG_RCAP regions, reads longer than RG_QCAP or long enough for the seed-SW filter (memchain.c:544),
// and two chains starting at the same reference position (there the reference's B-tree shape decides).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include "dev_common.hpp"
#include "wave.hpp"
#include "parsort.hpp"
#include "kernels.h"
#include "tune.h"
#include "ext_dp.hpp"
#include "rgx.hpp"

#define RG_QCAP 256
#define RG_NC 4          // extension rows live in registers: 64 * RG_NC entries >= qlen + 1

struct RgChain {         // mem_chain_t reduced to what chaining, the filter and the region loop read: first seed = (pos, first_q), last seed,
                         // and mem_chain_weight's two running sums (query and reference coverage) with their high-water marks
	long long pos, last_r;
	int rid, endr_off;                                        // endr_off: end of the reference coverage so far, relative to pos
	short first_q, last_q, last_len, wq, wr, endq, first, w;  // w = min(wq, wr) once chaining is over; first: mem_chain_flt's
	unsigned short n_seeds, seed0;                            // seeds on the main list (not seeds_extra), and the first of them
	signed char kept; unsigned char is_alt; unsigned short n_extra;   // n_extra: seeds on seeds_extra (contained in the chain on both axes)
};

// Working set of one strand search.  Two sizes: the common case lives in LDS; what does not fit there is
// redone by a second launch whose tables are a per-wave slab in HBM.
// B-tree node of the chain index (kbtree.h with t = 3: up to 5 keys); keys are chain ids, compared through their start positions
struct RgNode { unsigned char n, internal; unsigned short id[5], child[6]; };

// PCAP_ > 0 (the LDS tiers): only chains with more than one seed (or with contained seeds) have a record, at most PCAP_ of them; a chain
// of one seed IS that seed (every field of its record follows from it).  Against an hg38-sized genome a strand search has ~100 chains and
// all but a handful are such: the chain table was half of a wave's LDS, and LDS is what bounds these tiers' occupancy.
// Chain ids: with PCAP_ == 0 the record index; else < RG_REC the seed index of a one-seed chain, >= RG_REC record id - RG_REC.
#define RG_REC 0x4000
template <int ICAP_, int SCAP_, int CCAP_, int RCAP_, int NODES_, typename Idx, typename SIdx, int PCAP_ = 0>
struct RgStore {
	static constexpr int ICAP = ICAP_, SCAP = SCAP_, CCAP = CCAP_, RCAP = RCAP_, NODES = NODES_, PCAP = PCAP_;
	typedef Idx idx_t;
	// intervals (sorted by info): read until the last occurrence has been chained; the sort keys of the chains and of a chain's seeds
	// (srt) are first written after that, so the two share their bytes (2 KB of the 13.7 KB a wave of the larger LDS tier had: a
	// sixth workgroup per CU)
	union {
		struct { unsigned long long iv_x0[ICAP_]; int iv_n[ICAP_]; short iv_beg[ICAP_], iv_end[ICAP_];
		         unsigned long long iv_rank[NODES_ ? ICAP_ : 1]; };   // iv_n: count | more beyond << 29 | over-represented << 30; iv_rank (HBM tiers): first SA rank
		unsigned long long srt[SCAP_];    // score<<32|i, ascending (memchain.c:748-752)
	};
	// seeds in arrival order
	long long s_rbeg[SCAP_];
	int s_rid[SCAP_];
	short s_qbeg[SCAP_], s_len[SCAP_]; SIdx s_chain[SCAP_]; signed char s_extra[SCAP_];
	RgChain ch[PCAP_ ? PCAP_ : CCAP_];
	Idx ord[CCAP_];                   // chain indices: by position, then in filter order
	Idx keep[CCAP_];                  // mem_chain_flt's kept list (indices into ord)
	Idx lst[SCAP_];                   // seed indices of the current chain / list
	bsx_region_t regs[RCAP_ ? RCAP_ : 1];
	int n_chains, n_regs;
	int boost_iv;                     // status 10: the interval (sorted index) that has to be walked further
	// NODES > 0: chains are indexed by the reference's B-tree, so that chains starting at the same position are found
	// and ordered as kb_intervalp / __kb_traverse would (the first tier declines such tasks instead)
	RgNode node[NODES_ ? NODES_ : 1];
	int n_nodes, root;
};
typedef RgStore<64, 128, 128, 0, 0, unsigned short, short, 32> RgSmall;
typedef RgStore<128, 256, 256, 0, 0, unsigned short, short, 64> RgMid;             // still LDS: 24 KB per wave, two waves per workgroup: the
                                                                                // strand search of a read against an hg38-sized index (~50 intervals, ~125 seeds, ~100 chains)
// chunks with long reads (a kilobase against an hg38-sized index: ~360 intervals, ~830 seeds, most of them alone in their piece): still
// LDS, 62 KB per wave, two workgroups of one wave per CU -- every table access of the HBM tiers below is a memory round trip
typedef RgStore<640, 1152, 1152, 0, 0, unsigned short, short, 160> RgLongS;   // 49 KB: three workgroups per CU; four fifths of the kilobase reads' strand searches fit
typedef RgStore<768, 1536, 1536, 0, 0, unsigned short, short, 288> RgLongB;   // 67 KB, two per CU: most of the rest (288 chain records since round 6: 192 sent 6.7 k strand searches a chunk on to the HBM tier for their chains of several seeds, five times the time each: its launch 110 -> 43 ms, this one's 488 -> 522)
// ordinary reads inside repeat families (an hg38-like genome: 7 % of the strand searches outgrow RgMid): 23 KB per wave, six workgroups of one wave per CU
typedef RgStore<256, 512, 512, 0, 0, unsigned short, short, 96> RgMid2;
typedef RgStore<512, 1024, 1024, 1024, 1024, unsigned short, short> RgBig;
// the same capacity chained as the LDS tiers do it (pieces, chain starts in registers, records for multi-seed chains only), tables in an HBM
// slab, exporting: for the repeat reads that outgrow the LDS tiers but have no tied chain starts
typedef RgStore<4096, 8192, 8192, 8192, 8192, unsigned short, short> RgHuge;   // reads inside tandem repeats: thousands of short seeds
#define RG_WIN 768       // reference window of a chain kept in LDS while its seeds are extended (longer windows: extension reads HBM)
// $BSX_PHASES: where a strand search's wave cycles go.  The sums live in the wave's LDS and every lane writes them alike (uniform values): no
// divergent region inside the stages.  Round 5: `if (P.prof) { ...; if (lane == 0) atomicAdd(&counters[..], ..); }` ahead of the (not inlined)
// call of rg_export made the compiler place a copy of the lane index for the call AHEAD of the s_or_b64 that restores EXEC at the end of the
// lane-0 region -- 63 lanes entered rg_export with a stale lane index and exported garbage (tools/dbg/exec_join_check.py looks for that shape
// in the assembly; tests/test_isa_exec_join.py).  Flushed to counters[32 + k] once, when the wave leaves the kernel.
#define RG_NPF 16            // 0-7 stage cycles, 8 extensions, 9 extension rows
#define RG_PF_ZERO(D) do { if (P.prof) { (D).pf[lane & (RG_NPF - 1)] = 0; WAVE_SYNC(); } } while (0)
#define RG_PF_ADD(D, k, v) do { const unsigned long long s_ = (D).pf[k] + (unsigned long long)(v); (D).pf[k] = s_; } while (0)
#define RG_PF_FLUSH(D) do { if (P.prof) { WAVE_SYNC(); if (lane < RG_NPF) { const unsigned long long v_ = (D).pf[lane]; if (v_) atomicAdd(&counters[32 + lane], v_); } } } while (0)
struct RgDp {            // per-wave LDS scratch
	static const int QCAP = RG_QCAP;
	unsigned long long pf[RG_NPF];
	int32_t H[64], E[64];        // the one-lane passes (introsort stack, tree traversal stack)
	uint8_t q[RG_QCAP];          // the read
	uint8_t win[RG_WIN];         // reference bases [rmax0, rmax1) of the chain being extended, one byte each
};
struct RgDpLite { static const int QCAP = RG_QCAP; unsigned long long pf[RG_NPF]; int32_t H[64], E[64]; uint8_t q[RG_QCAP]; uint8_t win[4]; };   // the LDS tiers stop before the extensions: no window
#define RG_QCAP_LONG 1024   // the launches for chunks with longer reads (a read of a kilobase; anything longer is chained by the caller)
struct RgDpLiteL { static const int QCAP = RG_QCAP_LONG; unsigned long long pf[RG_NPF]; int32_t H[64], E[64]; uint8_t q[RG_QCAP_LONG]; uint8_t win[4]; };

// wave-uniform values live in scalar registers: say so for what comes out of LDS, shuffles and reductions
__device__ __forceinline__ long long uni64(long long v)
{
	return (long long)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

__device__ __forceinline__ long long wave_max_i64(long long v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const long long o = (long long)((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), off) << 32 | (unsigned)__shfl_xor((int)v, off));
		v = v > o ? v : o;
	}
	return v;
}

#define RG_CTG_LDS 128   // contig offset tables up to this many entries are copied to LDS once per workgroup
__device__ __forceinline__ int rg_pos2rid(const DevIndex &ix, const long long *ctg, long long pos_f)   // bns_pos2rid, bntseq.c:356-369
{
	int left = 0, mid = 0, right = ix.n_seqs;
	if (pos_f >= ix.l_pac) return -1;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ctg[mid]) {
			if (mid == ix.n_seqs - 1) break;
			if (pos_f < ctg[mid + 1]) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
__device__ __forceinline__ long long rg_depos(long long l_pac, long long p) { return p >= l_pac ? (l_pac << 1) - 1 - p : p; }
__device__ __forceinline__ int rg_intv2rid(const DevIndex &ix, const long long *ctg, long long rb, long long re)   // bns_intv2rid, bntseq.c:371-379
{
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	const int a = rg_pos2rid(ix, ctg, rg_depos(ix.l_pac, rb));
	const int b = rb < re ? rg_pos2rid(ix, ctg, rg_depos(ix.l_pac, re - 1)) : a;
	return a == b ? a : -1;
}
#define RG_BSS(parent, l_pac, rb) ((((rb) > (l_pac)) == (parent)) ? 1 : 0)

// klib introsort (ksort.h:184-236) of the chains by weight, descending, with the control flow of
// csrc/host/util.c:bsx_introsort so that equal weights end in the reference's order.  One lane runs it; the elements are
// packed keys (weight << RG_KEY_BITS | chain index) so that a comparison is two independent LDS reads and a swap two stores.
// asymmetric_flt_seed (memchain.c:138-149) for the seed of `ln` bases at reference position rb (forward-reverse space) whose read bases are q[0, ln): a
// reference T under a read C or a reference A under a read G.  The reference bases come 32 at a time (one unaligned 8-byte load of pac: a chance
// match of 19-22 bases is one load); a seed lies on one strand, the reverse one read backwards and complemented
__device__ __forceinline__ int rg_seed_conv_test(const DevIndex &ix, long long rb, int ln, const uint8_t *q)
{
	const long long l_pac = ix.l_pac;
	int bad = 0;
	const bool rev = rb >= l_pac;
	const long long f0 = rev ? (l_pac << 1) - rb - ln : rb;   // forward coordinate of the seed's lowest base
	for (int done = 0; done < ln; ) {
		const long long f = f0 + done;
		unsigned long long w;
		__builtin_memcpy(&w, ix.pac + (f >> 2), 8);   // (pac is padded: upload_ref)
		const int k0 = (int)(f & 3);
		int m = 32 - k0; if (m > ln - done) m = ln - done;
		for (int k = k0; k < k0 + m; ++k) {
			const int b = (int)(w >> (((k >> 2) << 3) + ((~k & 3) << 1))) & 3;
			const int at = done + (k - k0);   // position along the forward strand
			const int i = rev ? ln - 1 - at : at, r = rev ? 3 - b : b, qq = q[i];
			bad |= (r == 3 && qq == 1) || (r == 0 && qq == 2);
		}
		done += m;
	}
	return bad;
}

__device__ void rg_introsort_keys(unsigned int *a, int n, int *stk)
{
#define LT(x, y) (((x) >> RG_KEY_BITS) > ((y) >> RG_KEY_BITS))
#define SWP(i, j) do { const unsigned int t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
	if (n < 2) return;
	if (n == 2) { if (LT(a[1], a[0])) SWP(0, 1); return; }
	int d, s = 0, t = n - 1, i, j, k, top = 0;
	int *stk_l = stk, *stk_r = stk + 16, *stk_d = stk + 32;   // the longer side is stacked, and only when longer than 16: depth <= log2(n)
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { // comb sort fallback (ksort.h:162-183)
				const double shrink = 1.2473309501039786540366528676643;
				int m = t - s + 1, gap = m, swapped;
				do {
					if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
					swapped = 0;
					for (i = 0; i + gap < m; ++i) if (LT(a[s + i + gap], a[s + i])) { SWP(s + i, s + i + gap); swapped = 1; }
				} while (swapped || gap > 2);
				if (gap != 1) for (i = s + 1; i <= t; ++i) for (j = i; j > s && LT(a[j], a[j - 1]); --j) SWP(j, j - 1);
				t = s;
				continue;
			}
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (LT(a[k], a[i])) { if (LT(a[k], a[j])) k = j; }
			else k = LT(a[j], a[i]) ? i : j;
			const unsigned int rp = a[k];
			if (k != t) SWP(k, t);
			for (;;) {
				do ++i; while (LT(a[i], rp));
				do --j; while (i <= j && LT(rp, a[j]));
				if (j <= i) break;
				SWP(i, j);
			}
			SWP(i, t);
			if (i - s > t - i) {
				if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { for (i = 1; i < n; ++i) for (j = i; j > 0 && LT(a[j], a[j - 1]); --j) SWP(j, j - 1); return; }
			--top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
		}
	}
#undef LT
#undef SWP
}

// ---- the chain index as the reference keeps it: kbtree.h instantiated with t = 3 (pre-emptive split on the way down,
// lower-bound search inside a node, duplicates allowed); same structure as csrc/host/util.c:bsx_bt_*.  One lane runs it.
template <typename Store>
__device__ int rg_bt_find(const Store &S, const RgNode &x, long long pos, int &r)
{
	int begin = 0, end = x.n;
	r = 0;
	if (x.n == 0) return -1;
	while (begin < end) { const int mid = (begin + end) >> 1; if (S.ch[x.id[mid]].pos < pos) begin = mid + 1; else end = mid; }
	if (begin == x.n) { r = 1; return x.n - 1; }
	const long long kb = S.ch[x.id[begin]].pos;
	r = (kb < pos) - (pos < kb);
	if (r < 0) --begin;
	return begin;
}
template <typename Store>
__device__ int rg_bt_alloc(Store &S, int internal)
{
	RgNode &x = S.node[S.n_nodes];
	x.n = 0; x.internal = (unsigned char)internal;
	return S.n_nodes++;
}
template <typename Store>
__device__ void rg_bt_split(Store &S, int xi, int i, int yi)
{
	const int zi = rg_bt_alloc(S, S.node[yi].internal);
	RgNode &x = S.node[xi], &y = S.node[yi], &z = S.node[zi];
	z.n = 2;
	z.id[0] = y.id[3]; z.id[1] = y.id[4];
	if (y.internal) { z.child[0] = y.child[3]; z.child[1] = y.child[4]; z.child[2] = y.child[5]; }
	y.n = 2;
	for (int k = x.n; k > i; --k) x.child[k + 1] = x.child[k];
	x.child[i + 1] = (unsigned short)zi;
	for (int k = x.n; k > i; --k) x.id[k] = x.id[k - 1];
	x.id[i] = y.id[2];
	++x.n;
}
template <typename Store>
__device__ void rg_bt_put(Store &S, long long pos, int id)
{
	if (S.node[S.root].n == 5) {
		const int r = S.root, s = rg_bt_alloc(S, 1);
		S.node[s].child[0] = (unsigned short)r;
		S.root = s;
		rg_bt_split(S, s, 0, r);
	}
	int xi = S.root, r;
	for (;;) {
		RgNode &x = S.node[xi];
		if (!x.internal) {
			const int i = rg_bt_find(S, x, pos, r);
			for (int k = x.n - 1; k > i; --k) x.id[k + 1] = x.id[k];
			x.id[i + 1] = (unsigned short)id;
			++x.n;
			return;
		}
		int i = rg_bt_find(S, x, pos, r) + 1;
		if (S.node[x.child[i]].n == 5) {
			rg_bt_split(S, xi, i, x.child[i]);
			if (pos > S.ch[x.id[i]].pos) ++i;
		}
		xi = x.child[i];
	}
}
template <typename Store>
__device__ int rg_bt_lower(const Store &S, long long pos)
{
	int xi = S.root, lower = -1, r;
	for (;;) {
		const RgNode &x = S.node[xi];
		const int i = rg_bt_find(S, x, pos, r);
		if (i >= 0 && r == 0) return x.id[i];
		if (i >= 0) lower = x.id[i];
		if (!x.internal) return lower;
		xi = x.child[i + 1];
	}
}
// in-order traversal (__kb_traverse, kbtree.h:340-366) into ord[]
template <typename Store>
__device__ int rg_bt_traverse(Store &S, int *stk)
{
	int n = 0, top = 0;
	int *st_x = stk, *st_i = stk + 32;
	st_x[0] = S.root; st_i[0] = 0;
	while (top >= 0) {
		const RgNode &x = S.node[st_x[top]];
		const int i = st_i[top];
		if (!x.internal) { for (int k = 0; k < x.n; ++k) S.ord[n++] = (typename Store::idx_t)x.id[k]; --top; continue; }
		if (i > x.n) { --top; continue; }
		if (i > 0) S.ord[n++] = (typename Store::idx_t)x.id[i - 1];
		st_i[top] = i + 1;
		++top; st_x[top] = x.child[i]; st_i[top] = 0;
	}
	return n;
}

// bwt_sa (bwt.c:87-97) on one strand's own index
__device__ __forceinline__ long long rg_sa(const DevIndex &ix, int parent, unsigned long long k, uint32_t &n_steps)
{
	const unsigned long long prim = dev_ix_primary(ix, parent);
	const uint32_t *bw = dev_ix_bwt(ix, parent);
	const uint64_t *sa = parent ? ix.fmi[1].sa : ix.fmi[0].sa;
	const uint32_t sa_mask = ix.fmi[0].sa_mask, sa_shift = ix.fmi[0].sa_shift;
	unsigned long long steps = 0;
	while (k & sa_mask) {
		if (k == prim) { k = 0; ++steps; continue; }
		const unsigned long long x = k - (k > prim);
		const DevBlock B = dev_load_block4(bw, x);
		const int c = dev_planes_symbol(B, (int)(x & 127));
		uint32_t ca, cc, cg, ct;
		dev_planes_count4(B, (int)(x & 127), ca, cc, cg, ct);
		const unsigned long long base = c == 0 ? ((unsigned long long)B.v0.y << 32 | B.v0.x) : c == 1 ? ((unsigned long long)B.v0.w << 32 | B.v0.z) :
		                                c == 2 ? ((unsigned long long)B.v1.y << 32 | B.v1.x) : ((unsigned long long)B.v1.w << 32 | B.v1.z);
		k = dev_ix_L2(ix, parent, c) + base + (c == 0 ? ca : c == 1 ? cc : c == 2 ? cg : ct);
		++steps;
	}
	n_steps += (uint32_t)steps;
	return (long long)(steps + sa[k >> sa_shift]);
}

// overlap test of mem_chain_flt (memchain.c:436-452) for chain ci against kept chain ck: bit 0 = large overlap, bit 1 = ci is dropped
__device__ __forceinline__ int rg_flt_vals(const RegParams &P, int ci_beg, int ci_end, int ci_w, int ci_alt, int ck_beg, int ck_end, int ck_w, int ck_alt)
{
	const int b_max = ck_beg > ci_beg ? ck_beg : ci_beg, e_min = ck_end < ci_end ? ck_end : ci_end;
	if (e_min > b_max && (!ck_alt || ci_alt)) {
		const int li = ci_end - ci_beg, lj = ck_end - ck_beg, min_l = li < lj ? li : lj;
		if ((float)(e_min - b_max) >= (float)min_l * P.mask_level && min_l < P.max_chain_gap)
			return 1 | (((float)ci_w < (float)ck_w * P.drop_ratio && ck_w - ci_w >= P.min_seed_len << 1) ? 2 : 0);
	}
	return 0;
}
__device__ __forceinline__ int rg_flt_test(const RegParams &P, const RgChain &ci, const RgChain &ck)
{
	return rg_flt_vals(P, ci.first_q, ci.last_q + ci.last_len, ci.w, ci.is_alt, ck.first_q, ck.last_q + ck.last_len, ck.w, ck.is_alt);
}

// ---- The LDS tiers stop after the chain filter and hand the surviving chains to k_c2r: chaining needs its tables (24 KB of LDS per
// wave at the larger size, six waves per CU), the chain-to-region loop needs registers and latency hiding (extension rows are
// dependent DPP chains) but almost no tables.  One launch each, with the occupancy each can have.  What is exported per strand
// search: the kept chains in processing order, each with its seeds (main list, then seeds_extra) in arrival order.
// the record of chain `id` (by value); for a chain of one seed, made up from the seed (s_extra bit 2 = its contig is an ALT)
template <typename Store>
__device__ __forceinline__ RgChain rg_chain(const Store &S, int id)
{
	if (Store::PCAP == 0) return S.ch[id];
	if (id >= RG_REC) return S.ch[id - RG_REC];
	RgChain c;
	const short qb = S.s_qbeg[id], ln = S.s_len[id];
	c.pos = c.last_r = S.s_rbeg[id]; c.rid = S.s_rid[id]; c.endr_off = ln;
	c.first_q = c.last_q = qb; c.last_len = ln; c.wq = c.wr = ln; c.endq = (short)(qb + ln); c.first = -1; c.w = ln;
	c.n_seeds = 1; c.seed0 = (unsigned short)id; c.kept = 0; c.is_alt = (unsigned char)((S.s_extra[id] >> 2) & 1); c.n_extra = 0;
	return c;
}

// returns 11 (exported), 0 (nothing left to extend: the strand search has no regions) or 6 (no room: the next tier takes it)
template <typename Store>
__device__ int rg_export(Store &S, int t, int tot, float frac_rep, const RgXPool &X, int lane, int flt)
{
	typedef typename Store::idx_t idx_t;
	const int nk = uni(S.n_chains);
	const unsigned long long lt_mask = (1ull << lane) - 1;
	// a chain of one seed that fails asymmetric_flt_seed, with no contained seeds to fall back to, comes and goes without a trace
	int n_surv = 0, n_sd = 0;
	for (int base = 0; base < nk; base += 64) {
		const int ci = base + lane;
		bool surv = false; int cnt = 0;
		if (ci < nk) {
			const RgChain c = rg_chain(S, (int)S.ord[ci]);
			surv = !(c.n_seeds == 1 && c.n_extra == 0 && (S.s_extra[c.seed0] & 2));
			cnt = surv ? c.n_seeds + c.n_extra : 0;
		}
		const unsigned long long b = __ballot(surv);
		WAVE_SYNC();
		if (surv) S.ord[n_surv + __popcll(b & lt_mask)] = S.ord[ci];   // compacted in place, order kept (targets never pass the sources)
		n_surv += __popcll(b);
		n_sd += wave_sum_i32(cnt);
		WAVE_SYNC();
	}
	if (n_surv == 0) return 0;
	// room for the extensions made ahead of k_c2r (k_ext4.hip); the seed-SW filter rewrites the lists after the export.  X.ext = 1: a slot per chain (the
	// seed the loop reaches first); 2: a slot per seed (every seed of every main list is extended ahead: the HBM tiers' strand searches, where the
	// loop skips one seed in fourteen)
	const int ext = flt == RG_NOFLT ? X.ext : 0;
	const unsigned long long bytes = sizeof(RgXHdr) + (unsigned long long)n_surv * sizeof(RgXChain) + (unsigned long long)n_sd * sizeof(RgXSeed) + (ext ? (unsigned long long)(ext == 2 ? n_sd : n_surv) * sizeof(RgXExt) : 0ull);
	unsigned long long at = 0;
	if (lane == 0) at = atomicAdd(X.cursor, bytes);
	at = (unsigned long long)uni64((long long)at);
	if (at + bytes > X.cap) return 6;
	RgXHdr *H = (RgXHdr*)(X.base + at);
	RgXChain *XC = (RgXChain*)(H + 1);
	RgXSeed *XS = (RgXSeed*)(XC + n_surv);
	if (lane == 0) { H->n_chains = n_surv; H->n_seeds = n_sd; H->frac_rep = frac_rep; H->flt = flt; H->has_ext = ext; H->pad = Store::SCAP; }   // (pad: which tier's tables made the record, for the debug checks)
	int so = 0;
	for (int ci = 0; ci < n_surv; ++ci) {
		const int c = uni(S.ord[ci]);
		const RgChain chn = rg_chain(S, c);
		const int n_main = uni(chn.n_seeds), n_extra = uni(chn.n_extra);
		if (lane == 0) { RgXChain x; x.pos = chn.pos; x.rid = chn.rid; x.seed_off = so; x.n_main = (unsigned short)n_main; x.n_extra = (unsigned short)n_extra; x.pad = 0; XC[ci] = x; }
		if (n_main == 1 && n_extra == 0) {
			if (lane == 0) { const int o = chn.seed0; RgXSeed x; x.rbeg = S.s_rbeg[o]; x.qbeg = S.s_qbeg[o]; x.len = S.s_len[o]; x.sb = (int)x.len << 1 | ((S.s_extra[o] >> 1) & 1); XS[so] = x; }
			so += 1;
			continue;
		}
		for (int pass = 0; pass < 2; ++pass) {
			if (pass == 1 && n_extra == 0) break;
			int nl = 0;
			for (int base = 0; base < tot; base += 64) {
				const int o = base + lane;
				const bool in = o < tot && S.s_chain[o] == c && (int)(S.s_extra[o] & 1) == pass;
				const unsigned long long b = __ballot(in);
				if (in) { RgXSeed x; x.rbeg = S.s_rbeg[o]; x.qbeg = S.s_qbeg[o]; x.len = S.s_len[o]; x.sb = (int)x.len << 1 | ((S.s_extra[o] >> 1) & 1); XS[so + nl + __popcll(b & lt_mask)] = x; }
				nl += __popcll(b);
			}
			so += nl;
		}
	}
	if (lane == 0) { X.xoff[t] = (long long)at; X.xlist[atomicAdd(X.xcount, 1u)] = t; }
	return 11;
}

// One strand search, SA intervals -> regions, by one wavefront.  Returns 0 or the reason the task is declined:
//   1 seeding overflowed   9 read longer than RG_QCAP or long enough for the seed-SW filter (memchain.c:544)   8 intervals > ICAP
//   2 occurrences > SCAP or an interval beyond max_occ      3 chains > CCAP      4 two chains start at the same position
//   6 regions > RCAP      10 an over-represented interval has to be walked past max_occ (memchain.c:325-326)
template <typename Store, bool SPLIT = false, typename DP = RgDp>
__device__ int rg_task(Store &S, DP &D, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                       int l_query, int parent, uint32_t qoff, const DevIntv *src, int n_iv, const unsigned long long *posl, int lane,
                       unsigned long long *counters, const int *gap, const long long *ctg, const RgXPool *X = nullptr, int task_id = 0,
                       int n_boost = 0, const int *boost_iv = nullptr, const int *boost_lvl = nullptr)
{
	typedef typename Store::idx_t idx_t;
	const long long l_pac = ix.l_pac;
	const uint8_t *query = reads + qoff;
	// per-stage wave cycles (P.prof, set under $BSX_PHASES) into the wave's sums D.pf, see RG_PF_ADD
	long long pf_t = P.prof ? (long long)__builtin_readcyclecounter() : 0;
	unsigned int pf_ext = 0, pf_rows = 0;
#define RG_STAGE(k) do { if (P.prof) { const long long now_ = (long long)__builtin_readcyclecounter(); RG_PF_ADD(D, k, now_ - pf_t); pf_t = now_; } } while (0)
	if (lane == 0) {
		S.n_chains = 0; S.n_regs = 0;
		if (Store::NODES) { S.n_nodes = 0; S.root = rg_bt_alloc(S, 0); }
	}
	WAVE_SYNC();
	if (n_iv < 0) return 1;
	if (l_query > DP::QCAP) return 9;
	if (n_iv > Store::ICAP) return 8;
	// mem_flt_chained_seeds (memchain.c:537-568) runs for reads of this length?  Then the chains are exported with the threshold and
	// k_seedsw scores their seeds before k_c2r; the tiers that make regions themselves leave such a strand search to the caller
	const int flt = (P.flt_tab && l_query <= P.flt_len) ? uni(P.flt_tab[l_query]) : RG_NOFLT;
	if (!SPLIT && flt != RG_NOFLT) return 9;
	if (n_iv == 0 || l_query < P.min_seed_len) return 0;
	for (int i = lane; i < l_query; i += 64) D.q[i] = query[i];   // the read, for the seed tests and the extensions (ordered by the stage barriers below)

	// ---- A. intervals, ordered by info (ks_introsort(mem_intv), memchain.c:105; equal keys are identical records)
	// With positions looked up beforehand (k_occ), posl holds them interval by interval in the order the seeding kernel
	// left the list: iv_x0 then carries each interval's offset into posl instead of its first SA rank.
	long long run = 0;
	for (int base = 0; base < n_iv; base += 64) {
		const int i = base + lane;
		DevIntv mine; mine.x0 = mine.x1 = mine.x2 = 0; mine.info = 0;
		if (i < n_iv) mine = src[i];
		// occurrences this strand search will visit: all of them, or the first max_occ of an over-represented interval (memchain.c:325-326)
		const int big = mine.x2 > (unsigned long long)P.max_occ;
		const int lk = big ? P.max_occ : (int)mine.x2;   // what k_occ looked up ahead
		long long incl = lk;
		if (P.max_occ <= 1 << 24) incl = wave_scan_sum_incl(lk);   // (64 intervals of at most max_occ: the sum fits; DPP instead of twelve ds_bpermute)
		else {
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const long long o = (long long)((unsigned long long)(unsigned)__shfl_up((int)(incl >> 32), off) << 32 | (unsigned)__shfl_up((int)incl, off));
				if (lane >= off) incl += o;
			}
		}
		// the other keys come 64 at a time with one coalesced load and are handed round with v_readlane
		int rank = 0;
		constexpr bool K32 = Store::ICAP <= 1024 && DP::QCAP < 2048;   // begin (11 bits), end (11) and the list index (10) as ONE 32-bit key: unique, a readlane and a compare per pair
		if (K32) {
			const unsigned int mykey = i < n_iv ? ((unsigned int)(mine.info >> 32) << 21 | ((unsigned int)mine.info & 0x7ffu) << 10 | (unsigned int)i) : 0xffffffffu;
			for (int cb = 0; cb < n_iv; cb += 64) {
				const int kk = cb + lane;
				unsigned int okey = mykey;
				if (cb != base) { const unsigned long long oinfo = kk < n_iv ? src[kk].info : ~0ull; okey = kk < n_iv ? ((unsigned int)(oinfo >> 32) << 21 | ((unsigned int)oinfo & 0x7ffu) << 10 | (unsigned int)kk) : 0xffffffffu; }
				const int lim = n_iv - cb < 64 ? n_iv - cb : 64;
				for (int j = 0; j < lim; ++j) rank += (unsigned int)__builtin_amdgcn_readlane((int)okey, j) < mykey;
			}
		} else
		for (int cb = 0; cb < n_iv; cb += 64) {
			const int kk = cb + lane;
			const unsigned long long oinfo = kk < n_iv ? src[kk].info : ~0ull;
			const int lim = n_iv - cb < 64 ? n_iv - cb : 64;
			for (int j = 0; j < lim; ++j) {
				const unsigned long long oi = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(oinfo >> 32), j) << 32 | (unsigned)__builtin_amdgcn_readlane((int)oinfo, j);
				rank += (oi < mine.info) || (oi == mine.info && cb + j < i);
			}
		}
		if (i < n_iv) {
			// The reference walks an over-represented interval past its first max_occ occurrences while it has started at most five chains
			// (memchain.c:325-326).  The HBM tiers do too: when the rule asks for more of an interval the strand search is run again with that
			// interval's share multiplied (boost_*), the further positions walked here
			long long want = lk;
			if (Store::NODES && big) for (int b = 0; b < n_boost; ++b) if (boost_iv[b] == rank) want = (long long)P.max_occ * boost_lvl[b];
			if (want > (long long)mine.x2) want = (long long)mine.x2;
			if (want > 0x0fffffff) want = 0x0fffffff;
			const int cnt = (int)want;
			S.iv_x0[rank] = posl ? (unsigned long long)(run + incl - lk) : mine.x0;
			if (Store::NODES) S.iv_rank[rank] = mine.x0;
			S.iv_n[rank] = cnt | ((big && (unsigned long long)cnt < mine.x2) ? 1 << 29 : 0) | big << 30;
			S.iv_beg[rank] = (short)(mine.info >> 32); S.iv_end[rank] = (short)(uint32_t)mine.info;
		}
		run += uni64((long long)((unsigned long long)(unsigned)__shfl((int)(incl >> 32), 63) << 32 | (unsigned)__shfl((int)incl, 63)));
	}
	WAVE_SYNC();
	RG_STAGE(0);
	// ---- B. occurrences: every k < x[2] of every interval (the caps of memchain.c:325-326 cannot bind while x[2] <= max_occ)
	int tot = 0, over = 0, any_big = 0;
	for (int base = 0; base < n_iv; base += 64) { // totals with one pass over the table, 64 intervals at a time
		const int i = base + lane;
		const int v = i < n_iv ? S.iv_n[i] : 0;
		const int c = v & 0x1fffffff;
		if (__ballot(c > Store::SCAP)) over = 1;
		if (__ballot(v >> 30)) any_big = 1;
		tot += wave_sum_i32(c > Store::SCAP ? Store::SCAP : c);
		if (tot > Store::SCAP) { over = 1; break; }
	}
	if (over || tot > Store::SCAP) return 2;
	float frac_rep = 0.f;
	if (any_big) { // read length covered by over-represented seeds (memchain.c:294-301)
		int b = 0, e = 0, l_rep = 0;
		for (int i = 0; i < n_iv; ++i) {
			if (!(uni(S.iv_n[i]) >> 30)) continue;
			const int sb = uni(S.iv_beg[i]), se = uni(S.iv_end[i]);
			if (sb > e) { l_rep += e - b; b = sb; e = se; } else e = e > se ? e : se;
		}
		l_rep += e - b;
		frac_rep = (float)l_rep / l_query;
	}
	{
		int i = 0, acc = 0;   // occurrences are visited in increasing order by each lane: the interval cursor only moves forward
		uint32_t lf = 0;
		if (Store::SCAP > 256 && n_iv > 256) { // thousands of intervals with a few occurrences each: a lane per interval, 64 intervals at a time
			int run_o = 0;
			for (int base = 0; base < n_iv; base += 64) {
				const int ii = base + lane;
				const int cnt = ii < n_iv ? (S.iv_n[ii] & 0x1fffffff) : 0;
				const int incl = wave_scan_sum_incl(cnt);
				const int o0 = run_o + incl - cnt;
				if (cnt) {
					const unsigned long long x0 = S.iv_x0[ii];
					const int qb = S.iv_beg[ii], slen = S.iv_end[ii] - qb;
					const int lkc = (S.iv_n[ii] >> 30) && cnt > P.max_occ ? P.max_occ : cnt;   // looked up ahead; the rest is walked here
					const unsigned long long rk0 = Store::NODES ? S.iv_rank[ii] : x0;
					for (int c = 0; c < cnt; ++c) {
						const long long pos = (posl && c < lkc) ? (long long)posl[x0 + (unsigned long long)c] : rg_sa(ix, parent, (posl ? rk0 : x0) + (unsigned long long)c, lf);
						const int o = o0 + c;
						S.s_rbeg[o] = pos; S.s_qbeg[o] = (short)qb; S.s_len[o] = (short)slen;
						S.s_rid[o] = rg_intv2rid(ix, ctg, pos, pos + slen);
						S.s_chain[o] = -1; S.s_extra[o] = 0;
					}
				}
				run_o += uni(__builtin_amdgcn_readlane(incl, 63));
			}
		} else
		for (int o = lane; o < tot; o += 64) {
			while (acc + (S.iv_n[i] & 0x1fffffff) <= o) { acc += S.iv_n[i] & 0x1fffffff; ++i; }
			const int kk_ = o - acc, vv_ = S.iv_n[i], lkc_ = (vv_ >> 30) && (vv_ & 0x1fffffff) > P.max_occ ? P.max_occ : (vv_ & 0x1fffffff);
			const long long pos = (posl && kk_ < lkc_) ? (long long)posl[S.iv_x0[i] + (unsigned long long)kk_]
			                                            : rg_sa(ix, parent, ((posl && Store::NODES) ? S.iv_rank[Store::NODES ? i : 0] : S.iv_x0[i]) + (unsigned long long)kk_, lf);
			const int slen = S.iv_end[i] - S.iv_beg[i];
			S.s_rbeg[o] = pos; S.s_qbeg[o] = S.iv_beg[i]; S.s_len[o] = (short)slen;
			const int rid_ = rg_intv2rid(ix, ctg, pos, pos + slen);
			S.s_rid[o] = rid_;
			S.s_chain[o] = -1; S.s_extra[o] = (Store::PCAP && rid_ >= 0 && ix.ctg_alt[rid_]) ? 4 : 0;
		}
		// work counters of the algorithmic-bytes model: FM blocks touched by the LF walks, SA samples read
		lf = (uint32_t)wave_sum_i32((int)lf);
		if (lane == 0 && !posl) { atomicAdd(&counters[2], (unsigned long long)lf); atomicAdd(&counters[3], (unsigned long long)tot); }
	}
	WAVE_SYNC();
	// asymmetric_flt_seed (memchain.c:138-149) for every seed at once, a lane per seed: a reference T under a read C or a reference A
	// under a read G.  Bit 1 of s_extra; the seed loop of stage E only tests the bit (one dependent trip to HBM per seed there
	// was a quarter of this kernel's time on a genome where most seeds are chance matches).
	for (int o = lane; o < tot; o += 64) {
		if (S.s_rid[o] < 0) continue;
		const int bad = rg_seed_conv_test(ix, S.s_rbeg[o], (int)S.s_len[o], D.q + S.s_qbeg[o]);
		if (bad) S.s_extra[o] |= 2;
	}
	WAVE_SYNC();
	RG_STAGE(1);
	// ---- C. chaining in arrival order (mem_chain's loop over occurrences, memchain.c:313-366)
	// Without the B-tree (LDS tiers): the chain starts live sorted across the wave's registers, entry e in lane e & 63 of register set
	// e >> 6.  kb_intervalp's `lower` (the largest start <= rbeg) is a compare, a ballot and a population count per register set; a
	// new chain shifts the entries above it up one lane (DPP).  Unique starts make the sorted order the tree's in-order traversal.
	constexpr int TCAP = Store::CCAP < 512 ? Store::CCAP : 512;   // chain starts of one piece (the whole strand search in the HBM tiers' tree)
	constexpr int NR = Store::NODES ? 1 : (TCAP + 63) / 64;
	unsigned int st_lo[NR], st_hi[NR]; int st_id[NR];
#pragma unroll
	for (int r = 0; r < NR; ++r) { st_lo[r] = st_hi[r] = 0xffffffffu; st_id[r] = -1; }
	int nc = 0, n_prom = 0;   // chains; those of them with a record (PCAP stores)
	int cur_iv = -1, iv_stop = 0, iv_big = 0, count = 0;   // the interval the occurrence belongs to, and the chains it has started
	int iv_more = 0, iv_o0 = 0;                            // (HBM tiers) the interval has occurrences beyond the ones listed; its first seed
	// The LDS tiers first set aside the seeds that cannot meet any other: merge_seed_to_chain (memchain.c:227-256) only ever joins a
	// seed to a chain whose last seed lies less than l_query + min(w, max_chain_gap) before it on the reference (rdist <= qdist + w,
	// rdist - last->len < max_chain_gap) or whose span contains it, and every step inside a chain is that short too.  So cut the seeds,
	// sorted by reference position, wherever two neighbours are at least that far apart: a seed alone in its piece starts a chain of
	// its own whatever else the strand search holds, no later seed joins it, and it never changes the outcome of a test between two
	// other seeds (the chain below a seed is either in the seed's own piece or too far to be joined) -- against an hg38-sized genome
	// that is ~95 of the ~100 chains of a strand search, the chance matches of a 3-letter 19-mer.  Only the seeds of the other pieces
	// go through the sequential loop below, in arrival order, over a table of their own chain starts.
	int n_iso = 0, n_live = tot, n_pieces = 0;
	if (!Store::NODES) {
		constexpr int NS = Store::SCAP / 64;
		constexpr int KB = Store::SCAP <= 512 ? 9 : 12;   // bits of the arrival index below the position in the sort keys
		const long long dgap = (long long)l_query + 1 + (long long)(P.w < P.max_chain_gap ? P.w : P.max_chain_gap);
		unsigned long long key[NS]; int rk[NS];
		int n_dead = 0;
#pragma unroll
		for (int c = 0; c < NS; ++c) { // seeds that take no part (memchain.c:339-346) sort last
			const int o = (c << 6) + lane;
			key[c] = ~0ull; rk[c] = 0;
			bool dead = false;
			if (o < tot) {
				const long long rb = S.s_rbeg[o];
				dead = S.s_rid[o] < 0 || ((P.bsstrand & 1) && RG_BSS(parent, l_pac, rb) != P.bsstrand >> 1);
				if (dead) S.s_rbeg[o] = (1ll << 40) + o;   // nothing reads the position of such a seed again
				key[c] = (unsigned long long)(dead ? (1ll << 40) + o : rb) << KB | (unsigned)o;
			}
			n_dead += __popcll(__ballot(dead));
		}
		n_live = tot - n_dead;
		WAVE_SYNC();
		for (int k = 0; k < tot; ++k) { // rank by counting: the keys (position, arrival index) are unique
			const unsigned long long kk = (unsigned long long)uni64(S.s_rbeg[k]) << KB | (unsigned)k;
#pragma unroll
			for (int c = 0; c < NS; ++c) rk[c] += kk < key[c];
		}
#pragma unroll
		for (int c = 0; c < NS; ++c) { const int o = (c << 6) + lane; if (o < tot) S.lst[rk[c]] = (idx_t)o; }
		WAVE_SYNC();
		int iso = 0;
#pragma unroll
		for (int c = 0; c < NS; ++c) {
			const int r = (c << 6) + lane;
			bool opens = false; int o = 0;
			if (r < n_live) {
				o = S.lst[r];
				const long long rb = S.s_rbeg[o];
				const bool far_l = r == 0 || rb - S.s_rbeg[S.lst[r > 0 ? r - 1 : 0]] >= dgap;
				const bool far_r = r == n_live - 1 || S.s_rbeg[S.lst[r + 1 < n_live ? r + 1 : r]] - rb >= dgap;
				if (far_l && far_r) { S.s_chain[o] = (decltype(S.s_chain[0] + 0))o; S.s_extra[o] |= 8; ++iso; }   // bit 3: the seed starts a chain
				else { S.s_extra[o] |= 16; opens = far_l; }                                                          // bit 4: it goes through the loop
			}
			// the piece a seed with neighbours lies in (pieces numbered along the reference): keep[] is free until the chains are filtered
			const unsigned long long om = __ballot(opens);
			if (r < n_live && (S.s_extra[o] & 16)) S.keep[o] = (idx_t)(n_pieces + __popcll(om & ((2ull << lane) - 1)) - 1);
			n_pieces += __popcll(om);
		}
		n_iso = wave_sum_i32(iso);
		WAVE_SYNC();
	}
	// The pieces do not interact, so each is chained on its own, its seeds in arrival order over a table of its own chain starts: a read
	// inside a repeat family has hundreds of seeds in a hundred pieces of a few seeds each, and a search over one register set instead of
	// one per 64 chains of the whole strand search
	int o = -1, cbase = -64, piece = 0, nc_tot = 0;
	unsigned long long cmask = 0;
	for (;;) {
		if (Store::NODES) { // every occurrence, in arrival order
			if (++o >= tot) break;
			while (o >= iv_stop) { // next interval with occurrences
				// an over-represented interval is walked past its first max_occ occurrences while it has started at most 5 chains
				// (memchain.c:325-326): those further occurrences were not looked up, the host takes the strand search
				if (iv_more && count < P.max_occ && count <= 5) { if (lane == 0) S.boost_iv = cur_iv; return 10; }
				++cur_iv;
				const int v = uni(S.iv_n[cur_iv]);
				iv_o0 = iv_stop;
				iv_stop += v & 0x1fffffff; iv_big = v >> 30; iv_more = (v >> 29) & 1; count = 0;
			}
			// the loop condition of memchain.c:325-326: no more than max_occ chains from one interval; past the first max_occ occurrences only
			// while it has started at most five
			if (iv_big && (count >= P.max_occ || (count > 5 && o - iv_o0 >= P.max_occ))) { o = iv_stop - 1; continue; }
		} else { // the seeds of the current piece, in arrival order; then the next piece with an empty table
			while (cmask == 0 && piece < n_pieces) {
				cbase += 64;
				if (cbase >= tot) {
					++piece; cbase = -64;
					nc_tot += nc; nc = 0;
#pragma unroll
					for (int r = 0; r < NR; ++r) { st_lo[r] = st_hi[r] = 0xffffffffu; st_id[r] = -1; }
					continue;
				}
				const int oo = cbase + lane < tot ? cbase + lane : 0;
				cmask = __ballot(cbase + lane < tot && (S.s_extra[oo] & 16) && (int)S.keep[oo] == piece);
			}
			if (cmask == 0) break;
			o = cbase + (int)__builtin_ctzll(cmask);
			cmask &= cmask - 1;
		}
		const int rid = uni(S.s_rid[o]);
		if (rid < 0) continue;
		const long long rbeg = uni64(S.s_rbeg[o]);
		const int qbeg = uni(S.s_qbeg[o]), len = uni(S.s_len[o]);
		if (Store::NODES && (P.bsstrand & 1) && RG_BSS(parent, l_pac, rbeg) != P.bsstrand >> 1) continue;
		int lower = -1, at = 0;   // at: sorted index the new chain would take
		bool tied = false;
		if (Store::NODES) {
			if (lane == 0 && nc > 0) lower = rg_bt_lower(S, rbeg);
			lower = uni(__shfl(lower, 0));
		} else {
			const unsigned int rlo = (unsigned int)rbeg, rhi = (unsigned int)((unsigned long long)rbeg >> 32);
#pragma unroll
			for (int r = 0; r < NR; ++r)
				if (r * 64 < nc) at += __popcll(__ballot(st_hi[r] < rhi || (st_hi[r] == rhi && st_lo[r] <= rlo)));
			if (at > 0) {
				const int e = at - 1, el = e & 63;
				unsigned int plo = 0, phi = 0;
#pragma unroll
				for (int r = 0; r < NR; ++r)
					if (r == e >> 6) { lower = __builtin_amdgcn_readlane(st_id[r], el); plo = (unsigned int)__builtin_amdgcn_readlane((int)st_lo[r], el); phi = (unsigned int)__builtin_amdgcn_readlane((int)st_hi[r], el); }
				tied = plo == rlo && phi == rhi;
			}
		}
		int merged = 0;
		if (lower >= 0) { // merge_seed_to_chain, memchain.c:227-256
			const RgChain c = rg_chain(S, lower);
			int kind = 0;   // 1: contained in the chain (goes on seeds_extra), 2: appended
			if (rid == c.rid) {
				if (qbeg >= c.first_q && qbeg + len <= c.last_q + c.last_len && rbeg >= c.pos && rbeg + len <= c.last_r + c.last_len) kind = 1;
				else if (!((c.last_r < l_pac || c.pos < l_pac) && rbeg >= l_pac)) {
					const long long qdist = qbeg - c.last_q, rdist = rbeg - c.last_r;
					if (rdist >= 0 && qdist - rdist <= P.w && rdist - qdist <= P.w && qdist - c.last_len < P.max_chain_gap && rdist - c.last_len < P.max_chain_gap) kind = 2;
				}
			}
			kind = uni(kind);
			if (kind) {
				int rec = lower;   // where the chain's record is (made now if the chain was a single seed so far)
				if (Store::PCAP) {
					if (lower < RG_REC) {
						if (n_prom == Store::PCAP) return 3;
						rec = n_prom++;
						if (lane == 0) { S.ch[rec] = c; S.s_chain[c.seed0] = (decltype(S.s_chain[0] + 0))(RG_REC + rec); }
						const int e = at - 1, el = e & 63;   // its entry in the sorted table names the record from now on
#pragma unroll
						for (int r = 0; r < NR; ++r) if (r == e >> 6 && lane == el) st_id[r] = RG_REC + rec;
					} else rec = lower - RG_REC;
				}
				if (lane == 0) {
					RgChain &d = S.ch[rec];
					S.s_chain[o] = (decltype(S.s_chain[0] + 0))(Store::PCAP ? RG_REC + rec : rec);
					if (kind == 1) { S.s_extra[o] |= 1; d.n_extra = (unsigned short)(c.n_extra + 1); }
					else {
						d.last_q = (short)qbeg; d.last_r = rbeg; d.last_len = (short)len;
						// mem_chain_weight (memchain.c:158-180) as a running sum: seeds join a chain in the order that loop visits them
						int wq = c.wq, wr = c.wr;
						if (qbeg >= c.endq) wq += len; else if (qbeg + len > c.endq) wq += qbeg + len - c.endq;
						const long long endr = c.pos + c.endr_off;
						if (rbeg >= endr) wr += len; else if (rbeg + len > endr) wr += (int)(rbeg + len - endr);
						d.wq = (short)wq; d.wr = (short)(wr < 32767 ? wr : 32767);   // only min(wq, wr) is read, and wq <= read length
						if (qbeg + len > c.endq) d.endq = (short)(qbeg + len);
						if (rbeg + len > endr) d.endr_off = (int)(rbeg + len - c.pos);
						d.n_seeds = (unsigned short)(c.n_seeds + 1);
					}
				}
				merged = 1;
			}
		}
		if (!merged) {
			if (nc == (Store::NODES ? Store::CCAP : TCAP)) return 3;
			if (!Store::NODES && tied) return 4;   // duplicate key: the B-tree shape matters, the HBM tiers keep one
			const int new_id = Store::PCAP ? o : nc;   // a chain of one seed has no record in the LDS tiers
			if (lane == 0) {
				if (!Store::PCAP) {
					RgChain c;
					c.pos = c.last_r = rbeg; c.rid = rid; c.endr_off = len; c.first_q = c.last_q = (short)qbeg; c.last_len = (short)len;
					c.wq = c.wr = (short)len; c.endq = (short)(qbeg + len); c.first = -1; c.w = 0; c.n_seeds = 1; c.seed0 = (unsigned short)o;
					c.kept = 0; c.is_alt = ix.ctg_alt[rid] ? 1 : 0; c.n_extra = 0;
					S.ch[nc] = c;
				}
				S.s_chain[o] = (decltype(S.s_chain[0] + 0))new_id;
				if (!Store::NODES) S.s_extra[o] |= 8;
				if (Store::NODES) rg_bt_put(S, rbeg, nc);
			}
			if (!Store::NODES) { // open slot `at` of the sorted table
				const int ar = at >> 6, al = at & 63;
#pragma unroll
				for (int r = NR - 1; r >= 0; --r) {
					if (r >= ar && r * 64 <= nc) {
						const unsigned int e_lo = r > 0 ? (unsigned int)__builtin_amdgcn_readlane((int)st_lo[r > 0 ? r - 1 : 0], 63) : 0u;
						const unsigned int e_hi = r > 0 ? (unsigned int)__builtin_amdgcn_readlane((int)st_hi[r > 0 ? r - 1 : 0], 63) : 0u;
						const int e_id = r > 0 ? __builtin_amdgcn_readlane(st_id[r > 0 ? r - 1 : 0], 63) : 0;
						const unsigned int s_lo = (unsigned int)wave_prev((int)st_lo[r], (int)e_lo), s_hi = (unsigned int)wave_prev((int)st_hi[r], (int)e_hi);
						const int s_id = wave_prev(st_id[r], e_id);
						const bool keep = r == ar && lane < al, put = r == ar && lane == al;
						st_lo[r] = keep ? st_lo[r] : put ? (unsigned int)rbeg : s_lo;
						st_hi[r] = keep ? st_hi[r] : put ? (unsigned int)((unsigned long long)rbeg >> 32) : s_hi;
						st_id[r] = keep ? st_id[r] : put ? new_id : s_id;
					}
				}
			}
			++nc; ++count;
		}
		WAVE_SYNC();
	}
	if (Store::NODES && iv_more && count < P.max_occ && count <= 5) { if (lane == 0) S.boost_iv = cur_iv; return 10; }   // the last interval with occurrences, same rule
	const unsigned long long lt_mask_c = (1ull << lane) - 1;
	if (!Store::NODES) {
		WAVE_SYNC();
		if (any_big) { // the same rule: an over-represented interval that started at most 5 chains would be walked further
			int acc = 0;
			for (int i = 0; i < n_iv; ++i) {
				const int v = uni(S.iv_n[i]), cnt = v & 0x1fffffff;
				if (v >> 30) {
					int heads = 0;
					for (int b = acc; b < acc + cnt; b += 64) heads += __popcll(__ballot(b + lane < acc + cnt && (S.s_extra[b + lane < acc + cnt ? b + lane : acc] & 8)));
					if (heads < P.max_occ && heads <= 5) return 10;
				}
				acc += cnt;
			}
		}
		nc_tot += nc;
		if (nc_tot + n_iso > Store::CCAP) return 3;
		// chains in the order of their start positions (the in-order traversal of the reference's tree, memchain.c:372-379)
		int n_heads = 0;
		for (int base = 0; base < n_live; base += 64) {
			const int r = base + lane;
			bool hd = false; int id = 0;
			if (r < n_live) { const int oo = S.lst[r]; hd = (S.s_extra[oo] & 8) != 0; id = (int)S.s_chain[oo]; }
			const unsigned long long b = __ballot(hd);
			if (hd) S.ord[n_heads + __popcll(b & lt_mask_c)] = (idx_t)id;
			n_heads += __popcll(b);
		}
		nc = n_heads;
		WAVE_SYNC();
	}
	RG_STAGE(2);
	// ---- D. chain order = by start position; weights; filter (mem_chain_flt, memchain.c:406-488)
	if (nc > 0) {
		for (int c = lane; c < (Store::PCAP ? n_prom : nc); c += 64) { RgChain &d = S.ch[c]; const int w = d.wq < d.wr ? d.wq : d.wr; d.w = (short)w; }
		WAVE_SYNC();
		if (Store::NODES && lane == 0) rg_bt_traverse(S, D.E);
		WAVE_SYNC();
		RG_STAGE(3);
		// chains heavy enough, in that order, as sort keys; srt[] is free until stage E: first half keys, second half the kept list
		unsigned int *keys = (unsigned int*)S.srt, *keepc = keys + Store::SCAP;
		int n = 0;
		const unsigned long long lt_mask = (1ull << lane) - 1;
		for (int base = 0; base < nc; base += 64) {
			const int i = base + lane;
			const int c = i < nc ? (int)S.ord[i] : 0;
			const int w = i < nc ? (Store::PCAP ? (c >= RG_REC ? (int)S.ch[c - RG_REC].w : (int)S.s_len[c]) : (int)S.ch[c].w) : -1;
			const bool ok = i < nc && w >= P.min_chain_weight;
			const unsigned long long b = __ballot(ok);
			// the key carries the chain (its record index, or with PCAP its position in the order by start: ids do not fit the key there)
			if (ok) keys[n + __popcll(b & lt_mask)] = (unsigned int)w << RG_KEY_BITS | (unsigned int)(Store::PCAP ? i : c);
			if (Store::PCAP && i < nc) S.keep[i] = (idx_t)c;
			n += __popcll(b);
		}
		WAVE_SYNC();
		bool sorted = false;
		constexpr int SORT_MS = Store::CCAP <= 64 ? 1 : Store::CCAP <= 128 ? 2 : Store::CCAP <= 256 ? 4 : (Store::CCAP + 63) / 64;
		if (!Store::NODES && n <= 64 * SORT_MS) { // every partition at once (rg_introsort_par): the second half of srt[] holds the partners' lists
			// (the kilobase-read tiers: lst[] is free until the kept flags are written, E[] is the tree traversal's stack, which these tiers do not have)
			rg_introsort_par<SORT_MS>(keys, n, keys + Store::SCAP, keys + Store::SCAP + Store::SCAP / 2, D.H, lane, (short*)S.lst, D.E);
			sorted = true;
		}
		WAVE_SYNC();
		if (!sorted && lane == 0) rg_introsort_keys(keys, n, D.H);
		WAVE_SYNC();
		for (int i = lane; i < n; i += 64) { const unsigned int v = keys[i] & ((1u << RG_KEY_BITS) - 1); S.ord[i] = Store::PCAP ? S.keep[v] : (idx_t)v; }
		WAVE_SYNC();
		RG_STAGE(4);
		if (n > 0 && !Store::NODES) {
			// The overlap filter (mem_chain_flt, memchain.c:430-470) with the KEPT chains' numbers in registers: kept chain k lives in lane
			// k & 63 of register set k >> 6, in the order it was kept
			constexpr int NH = (Store::CCAP + 63) / 64;
			int kb[NH], kw[NH], kpos[NH], kst[NH], kfi[NH];
#pragma unroll
			for (int h = 0; h < NH; ++h) { kb[h] = kw[h] = 0; kpos[h] = -1; kst[h] = 0; kfi[h] = -1; }
			int n_kept = 1;
			{
				// A LANE PER CANDIDATE, 64 chains of the sorted order at a time, the kept list walked entry by entry (round 4): what chain i does to
				// a kept chain k, and k to i, depends on the two alone, and i meets the kept chains in the order they were kept.  So every lane
				// first tests its chain against the list as it stands (a drop ends the lane); then the lowest lane left is the next kept chain,
				// the lanes above it meet it straight from its lane's registers, and so on until no lane is left.  chn->first of a kept chain =
				// the lowest chain that overlaps it significantly having got that far = the lowest lane that says so in the one round it is met.
				// The test itself (memchain.c:436-452) in integers: "overlap >= min_l * mask_level" in float is overlap >= T(min_l) with
				// T(l) = ceil of the float product, T grows with l so T(min(li, lk)) = min(T(li), T(lk)); T >= 1 carries "e_min > b_max" and
				// T = 0xffff "min_l >= max_chain_gap"; "w_i < w_k * drop_ratio && w_k - w_i >= 2 * min_seed_len" is w_i < D(w_k).  T and D are
				// computed once per chain, a test is a max, two min, a subtraction and two compares, and who is alive, who overlapped somebody
				// and who is met are masks of the wave in scalar registers.
				// Kept chains: kb = begin | end << 16, kw = T | D << 16 | alt << 31, km = position | state << 12 | (first + 1) << 16.
				int km[NH];
#pragma unroll
				for (int h = 0; h < NH; ++h) km[h] = 0;
				const auto flt_T = [&](int l) -> int { if (!(l < P.max_chain_gap)) return 0xffff; const int t = (int)ceilf((float)l * P.mask_level); return t < 1 ? 1 : t; };
				const auto flt_D = [&](int w) -> int { int d = (int)ceilf((float)w * P.drop_ratio); const int e = w - (P.min_seed_len << 1) + 1; d = d < e ? d : e; return d < 0 ? 0 : d > 0x7fff ? 0x7fff : d; };
				// The chains no kept chain can drop are kept whatever the list holds: a drop needs w_i < w_k * drop_ratio and w_k - w_i >= 2 *
				// min_seed_len (memchain.c:449), no kept chain is heavier than the first, and the order is by weight -- so they are a PREFIX of the
				// sorted order (all of it for a strand search on the strand the read does not come from: a hundred chance matches of weight 19-22).
				// They enter the list up front, each from its own lane (entry = sorted position); what is left to find out about them -- whether
				// they overlap an earlier one significantly (kept = 2 instead of 3) and the first chain that does so to them -- are tests of
				// pairs that do not depend on each other: a pass over the list per 64 of them, no round per kept chain.
				int U = 0;
				{
					const int d0 = flt_D((int)(uni((int)keys[0]) >> RG_KEY_BITS));
					for (int base = 0; base < n; base += 64) {
						const int i = base + lane;
						const int wi = i < n ? (int)(keys[i] >> RG_KEY_BITS) : 0;
						const unsigned long long can = __ballot(i >= n || wi < d0);
						if (can) { U = base + (int)__builtin_ctzll(can); break; }
						U = base + 64;
					}
					if (U > n) U = n;
				}
#pragma unroll
				for (int h = 0; h < NH; ++h) {
					const int i = h * 64 + lane;
					if (h * 64 < U && i < U) {
						const RgChain cu = rg_chain(S, (int)S.ord[i]);
						const int b = cu.first_q, e = (int)cu.last_q + (int)cu.last_len;
						kb[h] = b | e << 16; kw[h] = flt_T(e - b) | flt_D((int)cu.w) << 16 | (int)((unsigned int)(cu.is_alt ? 1 : 0) << 31); km[h] = i | 3 << 12;
					}
				}
				n_kept = U;
				// When that prefix is the whole list nothing is left to find out: what the tests would add -- kept = 2 for 3, and the first
				// chain overlapping each (which mem_chain_flt turns into kept = 1, memchain.c:461-464) -- only ever tells kept chains apart
				// from each other, which nothing reads unless max_chain_extend is in force (memchain.c:467-475).  That is every strand search
				// on the strand its read does not come from: half of all, and the ones with the longest lists.
				const bool all_kept = U == n && P.max_chain_extend >= (unsigned int)n;
				for (int base = 0; base < n && !all_kept; base += 64) {
					const int i = base + lane;
					const bool valid = i >= 1 && i < n;
					int ib = 0, ie = 0, iw = 0, ia = 0;
					if (valid) { const RgChain ci = rg_chain(S, (int)S.ord[i]); ib = ci.first_q; ie = (int)ci.last_q + (int)ci.last_len; iw = ci.w; ia = ci.is_alt ? 1 : 0; }
					const int ti = flt_T(ie - ib);
					const int q0v = ib | ie << 16, q1v = ti | flt_D(iw) << 16 | (int)((unsigned int)ia << 31);
					unsigned long long live_m = __ballot(valid), large_m = 0;
					const unsigned long long pre_m = __ballot(i < U), alt_m = __ballot(ia != 0);   // pre: in the list already, meets the entries before its own
					const int k_end = base + 64 <= U ? base + 64 : n_kept;
					for (int k = 0; k < k_end; ++k) { // the list as it stands
						unsigned int p0 = 0, p1 = 0;
						const int kh = k >> 6, kl = k & 63;
#pragma unroll
						for (int h = 0; h < NH; ++h) if (h == kh) { p0 = (unsigned int)__builtin_amdgcn_readlane(kb[h], kl); p1 = (unsigned int)__builtin_amdgcn_readlane(kw[h], kl); }
						const int kbk = (int)(p0 & 0xffff), kek = (int)(p0 >> 16), tk = (int)(p1 & 0xffff), dk = (int)((p1 >> 16) & 0x7fff);
						const int ov = (kek < ie ? kek : ie) - (kbk > ib ? kbk : ib), tm = ti < tk ? ti : tk;
						unsigned long long hm = __ballot(ov >= tm) & live_m;
						if (k >= base) hm &= ~pre_m | (k - base < 63 ? ~0ull << (k - base + 1) : 0ull);
						if (p1 >> 31) hm &= alt_m;
						large_m |= hm;
						if (hm) {
							const int f = base + (int)__builtin_ctzll(hm);
#pragma unroll
							for (int h = 0; h < NH; ++h) if (h == kh) { if (lane == kl && !(km[h] >> 16)) km[h] |= (f + 1) << 16; }
							live_m &= ~(hm & __ballot(iw < dk));
						}
					}
					if (base < U) { // the entries of this batch's own lanes: kept = 2 when they overlap an earlier one
						const bool two = ((pre_m & live_m & large_m) >> lane) & 1;
#pragma unroll
						for (int h = 0; h < NH; ++h) if (h == base >> 6) { if (two) km[h] = (km[h] & ~(3 << 12)) | 2 << 12; }
					}
					unsigned long long um = live_m & ~pre_m;
					while (um) {
						const int l = (int)__builtin_ctzll(um);   // kept: the next entry of the list
						const unsigned int p0 = (unsigned int)__builtin_amdgcn_readlane(q0v, l), p1 = (unsigned int)__builtin_amdgcn_readlane(q1v, l);
						const int kbk = (int)(p0 & 0xffff), kek = (int)(p0 >> 16), tk = (int)(p1 & 0xffff), dk = (int)((p1 >> 16) & 0x7fff);
						const int ov = (kek < ie ? kek : ie) - (kbk > ib ? kbk : ib), tm = ti < tk ? ti : tk;
						unsigned long long hm = __ballot(ov >= tm) & live_m & (l < 63 ? ~0ull << (l + 1) : 0ull);
						if (p1 >> 31) hm &= alt_m;
						large_m |= hm;
						const unsigned long long dm = hm ? hm & __ballot(iw < dk) : 0ull;
						live_m &= ~dm;
						const int meta = (base + l) | (((large_m >> l) & 1) ? 2 : 3) << 12 | (hm ? base + (int)__builtin_ctzll(hm) + 1 : 0) << 16;
						const int sl = n_kept & 63, hi = n_kept >> 6;
#pragma unroll
						for (int h = 0; h < NH; ++h) if (h == hi) { if (lane == sl) { kb[h] = (int)p0; kw[h] = (int)p1; km[h] = meta; } }
						++n_kept;
						um &= ~dm & ~(1ull << l);
					}
				}
#pragma unroll
				for (int h = 0; h < NH; ++h) { kpos[h] = km[h] & 0xfff; kst[h] = (km[h] >> 12) & 3; kfi[h] = (km[h] >> 16) - 1; }
			}
			// kept flags by sorted position (PCAP: lst is free here), then the first chain each kept one shadows (chn->first, memchain.c:455-460)
			for (int j = lane; j < n; j += 64) { if (Store::PCAP) S.lst[j] = 0; else S.ch[S.ord[j]].kept = 0; }
			WAVE_SYNC();
#pragma unroll
			for (int h = 0; h < NH; ++h) if (lane + 64 * h < n_kept) { if (Store::PCAP) S.lst[kpos[h]] = (idx_t)kst[h]; else S.ch[S.ord[kpos[h]]].kept = (signed char)kst[h]; }
			WAVE_SYNC();
#pragma unroll
			for (int h = 0; h < NH; ++h) if (lane + 64 * h < n_kept && kfi[h] >= 0) { if (Store::PCAP) S.lst[kfi[h]] = 1; else S.ch[S.ord[kfi[h]]].kept = 1; }
			WAVE_SYNC();
		} else if (n > 0) {
			int nk = 1;
			if (lane == 0) { S.ch[S.ord[0]].kept = 3; S.keep[0] = 0; keepc[0] = S.ord[0]; }
			WAVE_SYNC();
			// no chain light enough for the heaviest one to drop it (memchain.c:449): then none is dropped by anybody, every chain is kept, and
			// the pair tests would only tell kept chains apart (2 for 3, `first`), which nothing reads unless max_chain_extend is in force
			bool all_kept = false;
			if (P.max_chain_extend >= (unsigned int)n) {
				const int w0 = (int)(uni((int)keys[0]) >> RG_KEY_BITS), wl = (int)(uni((int)keys[n - 1]) >> RG_KEY_BITS);
				all_kept = !((float)wl < (float)w0 * P.drop_ratio && w0 - wl >= P.min_seed_len << 1);
				if (all_kept) { for (int j = lane; j < n; j += 64) S.ch[S.ord[j]].kept = 3; WAVE_SYNC(); }
			}
			for (int i = 1; i < n && !all_kept; ++i) {
				// chain i against the kept chains, 64 at a time; the reference's loop stops at the first kept chain that drops it
				const int cidx = uni(S.ord[i]);
				const RgChain ci = S.ch[cidx];
				int stop = nk;
				for (int base = 0; base < nk && stop == nk; base += 64) {
					const int k = base + lane;
					const int r = k < nk ? rg_flt_test(P, ci, S.ch[keepc[k]]) : 0;
					const unsigned long long d = __ballot(r & 2);
					if (d) stop = base + __ffsll((long long)d) - 1;
				}
				int large = 0;
				for (int base = 0; base < nk && base <= stop; base += 64) {
					const int k = base + lane;
					int hit = 0;
					if (k < nk && k <= stop) { RgChain &ck = S.ch[keepc[k]]; hit = rg_flt_test(P, ci, ck) & 1; if (hit && ck.first < 0) ck.first = (short)i; }
					if (__ballot(hit)) large = 1;
				}
				if (stop == nk) {
					if (lane == 0) { S.keep[nk] = (idx_t)i; keepc[nk] = (unsigned int)cidx; S.ch[cidx].kept = large ? 2 : 3; }
					++nk;
				}
				WAVE_SYNC();
			}
			for (int k = lane; k < nk; k += 64) { const int f = S.ch[keepc[k]].first; if (f >= 0) S.ch[S.ord[f]].kept = 1; }
			WAVE_SYNC();
		}
		if (n > 0) {
			if (P.max_chain_extend < (unsigned int)n) { // at most max_chain_extend shadowed chains survive (memchain.c:474-482); off by default
				if (lane == 0) {
					int i; unsigned int k = 0;
					for (i = 0; i < n; ++i) { const int kp = Store::PCAP ? (int)S.lst[i] : (int)S.ch[S.ord[i]].kept; if (kp == 0 || kp == 3) continue; if (++k >= P.max_chain_extend) break; }
					for (; i < n; ++i) { if (Store::PCAP) { if (S.lst[i] < 3) S.lst[i] = 0; } else if (S.ch[S.ord[i]].kept < 3) S.ch[S.ord[i]].kept = 0; }
				}
				WAVE_SYNC();
			}
			int m = 0;   // surviving chains, in processing order, compacted in place (targets never pass the sources)
			for (int base = 0; base < n; base += 64) {
				const int i = base + lane;
				const int c = i < n ? (int)S.ord[i] : 0;
				const bool ok = i < n && (Store::PCAP ? S.lst[i] != 0 : S.ch[c].kept != 0);
				const unsigned long long b = __ballot(ok);
				WAVE_SYNC();
				if (ok) S.ord[m + __popcll(b & lt_mask)] = (idx_t)c;
				m += __popcll(b);
				WAVE_SYNC();
			}
			if (lane == 0) S.n_chains = m;
			WAVE_SYNC();
		}
	}
	RG_STAGE(5);
	if (SPLIT) { // the chain-to-region loop runs in k_c2r
		const int st = rg_export(S, task_id, tot, frac_rep, *X, lane, flt);
		RG_STAGE(6);
		return st;
	}
	// ---- E. chains -> regions (mem_chain2region, memchain.c:873-904)
	const int nk = uni(S.n_chains), ns = tot;
	unsigned int pf_seen = 0, pf_skip = 0, pf_inl = 0;
	for (int ci = 0; ci < nk; ++ci) {
		const int c = uni(S.ord[ci]);
		const RgChain chn = S.ch[c];
		const long long ch_pos = uni64(chn.pos);
		const int ch_has_extra = uni(chn.n_extra) != 0;
		const int single = uni(chn.n_seeds) == 1;
		const int seed0 = uni(chn.seed0);
		// a chain of one seed that fails asymmetric_flt_seed, with no contained seeds to fall back to, comes and goes without a trace:
		// most chains of a strand search against the wrong conversion are such
		if (single && !ch_has_extra && (uni(S.s_extra[seed0]) & 2)) continue;
		// mem_chain_reference_span (memchain.c:585-605) + bns_fetch_seq's contig clamp; one lane per seed
		long long rmax0 = l_pac << 1, rmax1 = 0;
		if (single) {
			const long long rb = uni64(S.s_rbeg[seed0]); const int qb = uni(S.s_qbeg[seed0]), ln = uni(S.s_len[seed0]);
			rmax0 = rb - (qb + rg_gap(gap, P, qb));
			rmax1 = rb + ln + ((l_query - qb - ln) + rg_gap(gap, P, l_query - qb - ln));
		} else {
			for (int o = lane; o < ns; o += 64) if (S.s_chain[o] == c && !(S.s_extra[o] & 1)) {
				const long long rb = S.s_rbeg[o]; const int qb = S.s_qbeg[o], ln = S.s_len[o];
				const long long b = rb - (qb + rg_gap(gap, P, qb));
				const long long e = rb + ln + ((l_query - qb - ln) + rg_gap(gap, P, l_query - qb - ln));
				rmax0 = rmax0 < b ? rmax0 : b; rmax1 = rmax1 > e ? rmax1 : e;
			}
			rmax0 = uni64(-wave_max_i64(-rmax0)); rmax1 = uni64(wave_max_i64(rmax1));
		}
		rmax0 = rmax0 > 0 ? rmax0 : 0; rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
		if (rmax0 < l_pac && l_pac < rmax1) { if (ch_pos < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
		int rid;
		{
			const int is_rev = ch_pos >= l_pac;
			rid = uni(chn.rid);   // a chain's seeds share the contig of its first one (memchain.c:232)
			long long far_beg = uni64(ctg[rid]), far_end = uni64(ctg[rid + 1]);
			if (is_rev) { const long long tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
			rmax0 = rmax0 > far_beg ? rmax0 : far_beg; rmax1 = rmax1 < far_end ? rmax1 : far_end;
		}
		const int n0 = uni(S.n_regs);
		int win_ok = 0;   // 0: not loaded yet, 1: in LDS, -1: too long for it
		for (int pass = 0; pass < 2; ++pass) {
			if (pass == 1 && !(uni(S.n_regs) == n0 && ch_has_extra)) break;
			// the list (seeds or seeds_extra) in arrival order, and its best-first order
			int nl = 0;
			if (single && pass == 0) {
				if (lane == 0) { S.lst[0] = (idx_t)seed0; S.srt[0] = (unsigned long long)(unsigned)S.s_len[seed0] << 32; }
				nl = 1;
				WAVE_SYNC();
			} else {
				for (int base = 0; base < ns; base += 64) {
					const int o = base + lane;
					const bool in = o < ns && S.s_chain[o] == c && (int)(S.s_extra[o] & 1) == pass;
					const unsigned long long b = __ballot(in);
					if (in) S.lst[nl + __popcll(b & ((1ull << lane) - 1))] = (idx_t)o;
					nl += __popcll(b);
				}
				WAVE_SYNC();
				for (int i = lane; i < nl; i += 64) { // keys score<<32|i are unique: rank by counting
					const unsigned long long key = (unsigned long long)(unsigned)S.s_len[S.lst[i]] << 32 | (unsigned)i;
					int r = 0;
					for (int k = 0; k < nl; ++k) r += ((unsigned long long)(unsigned)S.s_len[S.lst[k]] << 32 | (unsigned)k) < key;
					S.srt[r] = key;
				}
				WAVE_SYNC();
			}
			for (int k = nl - 1; k >= 0; --k) {
				const int si = uni((int)(uint32_t)S.srt[k]);
				const int o = uni(S.lst[si]);
				if (uni(S.s_extra[o]) & 2) continue;   // asymmetric_flt_seed (memchain.c:138-149), tested for every seed after stage B
				const long long s_rbeg = uni64(S.s_rbeg[o]); const int s_qbeg = uni(S.s_qbeg[o]), s_len = uni(S.s_len[o]);
				++pf_seen;
				// contained in a region of this strand search? (memchain.c:761-819)
				int u;
				const int nr = uni(S.n_regs);
				u = nr;
				for (int base = 0; base < nr && u == nr; base += 64) { // a lane per region made so far: the reference's loop stops at the first that passes
					bool hit = false;
					if (base + lane < nr) {
						const bsx_region_t &rg = S.regs[base + lane];
						if (!(s_rbeg < rg.rb || s_rbeg + s_len > rg.re || s_qbeg < rg.qb || s_qbeg + s_len > rg.qe) && !(s_len - rg.seedlen0 > .1 * l_query)) {
							int qd = s_qbeg - rg.qb; long long rd = s_rbeg - rg.rb;
							int max_gap = rg_gap(gap, P, (int)(qd < rd ? qd : rd));
							int w = max_gap < rg.w ? max_gap : rg.w;
							hit = qd - rd < w && rd - qd < w;
							qd = rg.qe - (s_qbeg + s_len); rd = rg.re - (s_rbeg + s_len);
							max_gap = rg_gap(gap, P, (int)(qd < rd ? qd : rd));
							w = max_gap < rg.w ? max_gap : rg.w;
							hit = hit || (qd - rd < w && rd - qd < w);
						}
					}
					const unsigned long long hm = __ballot(hit);
					if (hm) u = base + (int)__builtin_ctzll(hm);
				}
				if (u < nr) { // a later seed of the list (in sorted order) that may lead to a different alignment?  A lane per seed
					bool any = false;
					for (int base = k + 1; base < nl && !any; base += 64) {
						const int i = base + lane;
						bool alt = false;
						if (i < nl && S.srt[i] != 0) {
							const int oo = S.lst[(int)(uint32_t)S.srt[i]];
							const long long t_rbeg = S.s_rbeg[oo]; const int t_qbeg = S.s_qbeg[oo], t_len = S.s_len[oo];
							if (!(t_len < s_len * .95))
								alt = (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) ||
								      (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg);
						}
						any = __ballot(alt) != 0;
					}
					if (!any) { ++pf_skip; WAVE_SYNC(); if (lane == 0) S.srt[k] = 0; WAVE_SYNC(); continue; }
				}
				++pf_inl;
				// extension (memchain.c:613-730): left then right, each with up to MAX_BAND_TRY band widths
				if (win_ok == 0) { // the chain's reference window (bns_fetch_seq, memchain.c:889) into LDS, once, when a seed of it is first extended
					const int span = (int)(rmax1 - rmax0);
					if (span > 0 && span <= RG_WIN) {
						dev_fetch_window(D.win, ix.pac, l_pac, rmax0, span, lane);
						win_ok = 1;
						WAVE_SYNC();
					} else win_ok = -1;
				}
				bsx_region_t R; memset(&R, 0, sizeof(R));
				int aw0 = P.w, aw1 = P.w;
				const int qe = s_qbeg + s_len;
				R.score = R.truesc = -1; R.rid = rid;
				for (int side = 0; side < 2; ++side) {
					if (side == 0 && s_qbeg == 0) { R.score = R.truesc = s_len * P.a; R.qb = 0; R.rb = s_rbeg; continue; }
					if (side == 1 && qe == l_query) { R.qe = l_query; R.re = s_rbeg + s_len; continue; }
					const int sc0 = R.score, clip = side ? P.pen_clip3 : P.pen_clip5;
					int aw = P.w;
					bsx_ext_res_t res; res.score = -1; res.qle = res.tle = res.gtle = 0; res.gscore = -1; res.max_off = 0;
					bsx_ext_job_t J;
					J.parent = (uint8_t)parent; J.pad = 0; J.end_bonus = clip;
					if (side == 0) { J.qoff = qoff + (uint32_t)s_qbeg - 1; J.qdir = -1; J.qlen = s_qbeg; J.tpos = s_rbeg - 1; J.tdir = -1; J.tlen = (int)(s_rbeg - rmax0); J.h0 = s_len * P.a; }
					else { J.qoff = qoff + (uint32_t)qe; J.qdir = 1; J.qlen = l_query - qe; J.tpos = s_rbeg + s_len; J.tdir = 1; J.tlen = (int)(rmax1 - (s_rbeg + s_len)); J.h0 = sc0; }
					for (int i = 0; i < 2; ++i) {
						const int prev = R.score;
						aw = P.w << i;
						J.w = aw;
						RG_STAGE(6);
						// most extensions of a 150 bp read are shorter than a wavefront is wide: one register entry per lane then,
						// and none of the per-chunk band tests and carries of the wider form
						if (J.qlen < 64) res = ext_dp_reg<1>(ix, sc, reads, J, lane, win_ok > 0 ? D.win : nullptr, rmax0, D.q, qoff);
						else if (J.qlen < 128) res = ext_dp_reg<2>(ix, sc, reads, J, lane, win_ok > 0 ? D.win : nullptr, rmax0, D.q, qoff);
						else res = ext_dp_reg<RG_NC>(ix, sc, reads, J, lane, win_ok > 0 ? D.win : nullptr, rmax0, D.q, qoff);   // qlen <= l_query - 1 <= 255: fits the 256 register entries
						res.score = uni(res.score); res.qle = uni(res.qle); res.tle = uni(res.tle); res.gtle = uni(res.gtle);
						res.gscore = uni(res.gscore); res.max_off = uni(res.max_off);
						R.score = res.score;
						RG_STAGE(7);
						++pf_ext; pf_rows += (unsigned int)(res.tle > res.gtle ? res.tle : res.gtle);
						if (R.score == prev || res.max_off < (aw >> 1) + (aw >> 2)) break;
					}
					const int local = res.gscore <= 0 || res.gscore <= R.score - clip;
					if (side == 0) {
						aw0 = aw;
						if (local) { R.qb = s_qbeg - res.qle; R.rb = s_rbeg - res.tle; R.truesc = R.score; }
						else { R.qb = 0; R.rb = s_rbeg - res.gtle; R.truesc = res.gscore; }
					} else {
						aw1 = aw;
						if (local) { R.qe = qe + res.qle; R.re = s_rbeg + s_len + res.tle; R.truesc += R.score - sc0; }
						else { R.qe = l_query; R.re = s_rbeg + s_len + res.gtle; R.truesc += res.gscore - sc0; }
					}
				}
				R.bss = (uint8_t)RG_BSS(parent, l_pac, R.rb); R.parent = (uint8_t)parent;
				if (RG_BSS(parent, l_pac, R.re) != R.bss) continue;   // crosses the strand boundary (memchain.c:846-849)
				int cov = 0;
				for (int i = lane; i < nl; i += 64) {
					const int oo = S.lst[i];
					const long long t_rbeg = S.s_rbeg[oo]; const int t_qbeg = S.s_qbeg[oo], t_len = S.s_len[oo];
					if (t_qbeg >= R.qb && t_qbeg + t_len <= R.qe && t_rbeg >= R.rb && t_rbeg + t_len <= R.re) cov += t_len;
				}
				R.seedcov = uni(wave_sum_i32(cov));
				R.w = aw0 > aw1 ? aw0 : aw1; R.seedlen0 = s_len; R.frac_rep = frac_rep;
				if (uni(S.n_regs) == Store::RCAP) return 6;
				WAVE_SYNC();
				if (lane == 0) { S.regs[S.n_regs] = R; ++S.n_regs; }
				WAVE_SYNC();
			}
		}
	}
	RG_STAGE(6);
	if (P.prof) { RG_PF_ADD(D, 8, pf_ext); RG_PF_ADD(D, 9, pf_rows); RG_PF_ADD(D, 11, pf_seen); RG_PF_ADD(D, 12, pf_skip); RG_PF_ADD(D, 14, pf_inl); }
	return 0;
}

// publish the regions of task t (or the reason it was declined)
template <typename Store>
__device__ __forceinline__ int rg_publish(Store &S, int t, int status, bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor,
                                          long long *reg_off, int *reg_n, int lane)
{
	WAVE_SYNC();
	if (lane == 0) {
		const int n = status ? 0 : S.n_regs;
		unsigned long long base = 0;
		if (n > 0) {
			base = atomicAdd(out_cursor, (unsigned long long)n);
			if (base + n <= out_cap) for (int k = 0; k < n; ++k) out[base + k] = S.regs[k];
			else status = 7;
		}
		reg_off[t] = (long long)base;
		reg_n[t] = status ? -status : n;
	}
	status = uni(__shfl(status, 0));
	WAVE_SYNC();
	return status;
}

// ---- chains -> regions for the strand searches the LDS tiers exported (mem_chain2region + mem_chain2region1, memchain.c:742-904):
// the same seed loop as stage E of rg_task, over the exported lists.  One wavefront per strand search; nothing but the regions made
// so far, the read, the current chain's seeds and its reference window live in LDS (5.5 KB per wave), so the launch runs at the
// occupancy the registers allow.
#define RG_XSEEDS 128    // seeds of one list (main or seeds_extra) of a chain held in LDS; longer lists: the next tier takes the strand search
#define RG_XREGS 64      // regions of one strand search (a lane each in the containment test); a read inside a high-copy repeat has dozens
#define RG_XCBLK 16      // chain records staged at a time
template <int QC, int WC, int XSD, int XRG = RG_XREGS, bool ROWS_ = true>
struct RgC2rT {
	static const int QCAP = QC, WINCAP = WC, XSEEDS = XSD, XREGS = XRG, GAPCAP = QC > RG_QCAP ? RG_QCAP : QC;   // (cal_max_gap tabulated up to GAPCAP; LDS is what bounds the long-read launch)
	static const bool HBM = false, ROWS = ROWS_;   // ROWS: LDS for an extension's rows (ext_dp: bands of more than 255 columns of a long read); without them such a strand search is handed on
	unsigned long long pf[RG_NPF];
	bsx_region_t regs[XRG];
	RgXChain xc[RG_XCBLK];   // chains [xc_lo, xc_lo + RG_XCBLK) of the exported record
	RgXExt xe[RG_XCBLK];     // and, when the record has them, the extensions made ahead of this launch (k_ext4.hip)
	RgXSeed sd[XSD];         // a window [sd_lo, sd_hi) over the record's seeds: the current chain's lists lie inside it
	unsigned long long srt[XSD];
	uint8_t q[QC];
	uint8_t win[WC];
	int n_regs;
	// long reads: the extension's rows in LDS (ext_dp: registers for the band only, so that the launch keeps several waves per SIMD;
	// rows in registers would need a slot per 64 query bases, 16 of them for a kilobase)
	int32_t Hrow[(ROWS_ && QC > RG_QCAP) ? QC + 2 : 1], Erow[(ROWS_ && QC > RG_QCAP) ? QC + 2 : 1];
	uint8_t qrow[(ROWS_ && QC > RG_QCAP) ? QC : 4];
};
typedef RgC2rT<RG_QCAP, RG_WIN, RG_XSEEDS> RgC2r;
// (round 6: no LDS rows -- 9 KB of the wave's 22 -- since the extensions of a long read keep their rows in a register window, ext_dp_win; the few
// whose band outgrows it, a second band width of 401 columns, go on to k_c2r<RgC2rHL> with the strand search: twelve waves per CU instead of six)
typedef RgC2rT<RG_QCAP_LONG, 1536, 256, RG_XREGS, false> RgC2rL;
// The same workspace with its three large tables -- the regions made so far, the window over the record's seeds, the sort keys -- in a slab of
// HBM per wave (round 6): for the strand searches of reads inside repeat families that outgrow RgC2rB (up to 1024 regions, 1024 seeds a list: what
// the first HBM tier holds).  Until round 6 those went through that tier's monolithic form, whose extensions run inline, a wavefront each -- 58 % of
// the tier's cycles, a quarter of all wave cycles of a chunk; through this launch their chains' extensions come from k_extl / k_ext4 (several
// jobs to a wavefront, made ahead) and only the seed loop's bookkeeping pays HBM round trips.
template <int QC, int WC, int XSD, int XRG>
struct RgC2rHT {
	static const int QCAP = QC, WINCAP = WC, XSEEDS = XSD, XREGS = XRG, GAPCAP = QC > RG_QCAP ? RG_QCAP : QC;
	static const bool HBM = true, ROWS = true;
	unsigned long long pf[RG_NPF];
	bsx_region_t *regs;      // XRG entries in the wave's slab of HBM: written once per region, read a lane per region by the containment test
	RgXChain xc[RG_XCBLK];
	RgXExt xe[RG_XCBLK];
	RgXSeed sd[XSD];         // (the seed window and its sort keys stay in LDS: the rank by counting reads every key once per seed)
	unsigned long long srt[XSD];
	uint8_t q[QC];
	uint8_t win[WC];
	int n_regs;
	int32_t Hrow[QC > RG_QCAP ? QC + 2 : 1], Erow[QC > RG_QCAP ? QC + 2 : 1];   // (long reads: the extension's rows in LDS, as in RgC2rT)
	uint8_t qrow[QC > RG_QCAP ? QC : 4];
	static constexpr size_t slab_bytes() { return (size_t)XRG * sizeof(bsx_region_t); }
};
typedef RgC2rHT<RG_QCAP, RG_WIN, 1024, 1024> RgC2rH;   // 27 KB of LDS: five workgroups of one wave per CU
typedef RgC2rHT<RG_QCAP_LONG, 1536, 1024, 1024> RgC2rHL;   // reads up to a kilobase: what k_c2r<RgC2rL> declines (64 regions, 256 seeds a list) -- chained on the host until round 6
typedef RgC2rT<RG_QCAP, RG_WIN, 256, 256> RgC2rB;   // reads inside repeat families: up to 256 regions of a strand search, 256 seeds of a chain (22 KB: seven waves per CU); for what k_c2r<RgC2r> declines   // reads up to a kilobase: a chain's window is the read plus its two gaps, a true chain has a few hundred seeds

template <typename WT>
__device__ int rg_c2r(WT &W, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, int l_query, int parent, uint32_t qoff,
                      const RgXHdr *H, int lane, unsigned long long *counters, const int *gap, const long long *ctg)
{
	const long long l_pac = ix.l_pac;
	long long pf_t = P.prof ? (long long)__builtin_readcyclecounter() : 0;
	unsigned int pf_ext = 0, pf_rows = 0, pf_seen = 0, pf_skip = 0, pf_cached = 0, pf_inl = 0;
	// The exported record (header, chains in processing order, their seeds in the same order) is staged through LDS a block of
	// chains and a window of seeds at a time: two round trips to HBM per strand search (header + read, then chains + seeds) where
	// reading each chain's record and lists where they lie was three per chain, twelve chains per strand search.
	for (int i = lane; i < l_query; i += 64) W.q[i] = reads[qoff + i];
	const int nk = uni(H->n_chains), n_sd = uni(H->n_seeds);
	// routing, before anything is started: a strand search with more chains than this launch holds regions, or with a list longer than its seed
	// window, goes on to the launch with larger tables (it would be declined half-way otherwise, its work done twice)
	if (nk > 4 * WT::XREGS) return 6;
	{
		const RgXChain *XCh = (const RgXChain*)(H + 1);
		int mx = 0;
		for (int c = lane; c < nk; c += 64) { const int a = XCh[c].n_main, b = XCh[c].n_extra; mx = mx > a ? mx : a; mx = mx > b ? mx : b; }
		if (uni(wave_max_i32(mx)) > WT::XSEEDS) return 2;
	}
	const float frac_rep = H->frac_rep;
	const unsigned long long *XCw = (const unsigned long long*)(H + 1);                           // RgXChain = 3 words, RgXSeed = 2
	const unsigned long long *XSw = XCw + (size_t)nk * (sizeof(RgXChain) / 8);
	const int has_ext = uni(H->has_ext);
	const unsigned long long *XEw = XSw + (size_t)n_sd * (sizeof(RgXSeed) / 8);
	if (lane == 0) W.n_regs = 0;
	int xc_lo = 0, sd_lo = 0, sd_hi = 0;
#define C2R_STAGE_CHAINS(from) do { const int n_ = (nk - (from) < RG_XCBLK ? nk - (from) : RG_XCBLK) * (int)(sizeof(RgXChain) / 8); \
		for (int i_ = lane; i_ < n_; i_ += 64) ((unsigned long long*)W.xc)[i_] = XCw[(size_t)(from) * (sizeof(RgXChain) / 8) + i_]; \
		if (has_ext == 1) for (int i_ = lane; i_ < n_ * 2; i_ += 64) ((unsigned long long*)W.xe)[i_] = XEw[(size_t)(from) * (sizeof(RgXExt) / 8) + i_]; xc_lo = (from); } while (0)
#define C2R_STAGE_SEEDS(from) do { const int m_ = n_sd - (from) < WT::XSEEDS ? n_sd - (from) : WT::XSEEDS; \
		for (int i_ = lane; i_ < m_ * 2; i_ += 64) ((unsigned long long*)W.sd)[i_] = XSw[(size_t)(from) * 2 + i_]; sd_lo = (from); sd_hi = (from) + m_; } while (0)
	C2R_STAGE_CHAINS(0);
	C2R_STAGE_SEEDS(0);
	WAVE_SYNC();
	for (int ci = 0; ci < nk; ++ci) {
		if (ci >= xc_lo + RG_XCBLK) { WAVE_SYNC(); C2R_STAGE_CHAINS(ci); WAVE_SYNC(); }
		const RgXChain &XCc = W.xc[ci - xc_lo];
		const long long ch_pos = uni64(XCc.pos);
		const int rid = uni(XCc.rid), seed_off = uni(XCc.seed_off), n_main = uni((int)XCc.n_main), n_extra = uni((int)XCc.n_extra);
		if (n_main == 0) continue;   // every seed of the chain failed the seed-SW filter (memchain.c:880)
		if (n_main > WT::XSEEDS || n_extra > WT::XSEEDS) return 2;
		if (seed_off + n_main > sd_hi) { WAVE_SYNC(); C2R_STAGE_SEEDS(seed_off); WAVE_SYNC(); }
		// mem_chain_reference_span (memchain.c:585-605) over the main list + bns_fetch_seq's contig clamp
		long long rmax0 = l_pac << 1, rmax1 = 0;
		for (int o = lane; o < n_main; o += 64) {
			const RgXSeed sd = W.sd[seed_off - sd_lo + o];
			const long long b = sd.rbeg - (sd.qbeg + rg_gap(gap, P, sd.qbeg));
			const long long e = sd.rbeg + sd.len + ((l_query - sd.qbeg - sd.len) + rg_gap(gap, P, l_query - sd.qbeg - sd.len));
			rmax0 = rmax0 < b ? rmax0 : b; rmax1 = rmax1 > e ? rmax1 : e;
		}
		if (n_main == 1) { // most chains are one seed: lane 0 holds the span
			rmax0 = uni64(rmax0); rmax1 = uni64(rmax1);
		} else { rmax0 = uni64(-wave_max_i64(-rmax0)); rmax1 = uni64(wave_max_i64(rmax1)); }
		rmax0 = rmax0 > 0 ? rmax0 : 0; rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
		if (rmax0 < l_pac && l_pac < rmax1) { if (ch_pos < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
		{
			const int is_rev = ch_pos >= l_pac;
			long long far_beg = uni64(ctg[rid]), far_end = uni64(ctg[rid + 1]);
			if (is_rev) { const long long tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
			rmax0 = rmax0 > far_beg ? rmax0 : far_beg; rmax1 = rmax1 < far_end ? rmax1 : far_end;
		}
		const int n0 = uni(W.n_regs);
		int win_ok = 0;
		for (int pass = 0; pass < 2; ++pass) {
			if (pass == 1 && !(uni(W.n_regs) == n0 && n_extra > 0)) break;
			const int nl = pass ? n_extra : n_main;
			const int l0 = seed_off + (pass ? n_main : 0);
			if (l0 + nl > sd_hi) { WAVE_SYNC(); C2R_STAGE_SEEDS(l0); WAVE_SYNC(); }
			const RgXSeed *Lsd = W.sd + (l0 - sd_lo);   // this list, in place in the window
			for (int i = lane; i < nl; i += 64) { // keys score<<32|i are unique: rank by counting (ks_introsort_64, memchain.c:752)
				const unsigned long long key = (unsigned long long)(unsigned)XS_SCORE(Lsd[i]) << 32 | (unsigned)i;
				int r = 0;
				for (int k = 0; k < nl; ++k) r += ((unsigned long long)(unsigned)XS_SCORE(Lsd[k]) << 32 | (unsigned)k) < key;
				W.srt[r] = key;
			}
			WAVE_SYNC();
			for (int k = nl - 1; k >= 0; --k) {
				const int si = uni((int)(uint32_t)W.srt[k]);
				const RgXSeed sd = Lsd[si];
				if (uni(XS_BAD(sd))) continue;   // asymmetric_flt_seed (memchain.c:138-149), tested by the tier that exported the seed
				const long long s_rbeg = uni64(sd.rbeg); const int s_qbeg = uni((int)sd.qbeg), s_len = uni((int)sd.len);
				++pf_seen;
				// contained in a region of this strand search? (memchain.c:761-819)
				int u;
				const int nr = uni(W.n_regs);
				u = nr;
				for (int rbase = 0; rbase < nr && u == nr; rbase += 64) { // a lane per region made so far, 64 at a time: the reference's loop stops at the first region that passes
					bool hit = false;
					if (rbase + lane < nr) {
						const bsx_region_t &rg = W.regs[rbase + lane];
						if (!(s_rbeg < rg.rb || s_rbeg + s_len > rg.re || s_qbeg < rg.qb || s_qbeg + s_len > rg.qe) && !(s_len - rg.seedlen0 > .1 * l_query)) {
							int qd = s_qbeg - rg.qb; long long rd = s_rbeg - rg.rb;
							int max_gap = rg_gap(gap, P, (int)(qd < rd ? qd : rd));
							int w = max_gap < rg.w ? max_gap : rg.w;
							hit = qd - rd < w && rd - qd < w;
							qd = rg.qe - (s_qbeg + s_len); rd = rg.re - (s_rbeg + s_len);
							max_gap = rg_gap(gap, P, (int)(qd < rd ? qd : rd));
							w = max_gap < rg.w ? max_gap : rg.w;
							hit = hit || (qd - rd < w && rd - qd < w);
						}
					}
					const unsigned long long hm = __ballot(hit);
					if (hm) u = rbase + (int)__builtin_ctzll(hm);
				}
				if (u < nr) { // is there a later seed of the list (in sorted order) that may lead to a different alignment?  A lane per seed
					bool any = false;
					for (int base = k + 1; base < nl && !any; base += 64) {
						const int i = base + lane;
						bool alt = false;
						if (i < nl && W.srt[i] != 0) {
							const RgXSeed td = Lsd[(int)(uint32_t)W.srt[i]];
							const long long t_rbeg = td.rbeg; const int t_qbeg = td.qbeg, t_len = td.len;
							if (!(t_len < s_len * .95))
								alt = (s_qbeg <= t_qbeg && s_qbeg + s_len - t_qbeg >= s_len >> 2 && t_qbeg - s_qbeg != t_rbeg - s_rbeg) ||
								      (t_qbeg <= s_qbeg && t_qbeg + t_len - s_qbeg >= s_len >> 2 && s_qbeg - t_qbeg != s_rbeg - t_rbeg);
						}
						any = __ballot(alt) != 0;
					}
					if (!any) { ++pf_skip; WAVE_SYNC(); if (lane == 0) W.srt[k] = 0; WAVE_SYNC(); continue; }
				}
				// extension (memchain.c:613-730): left then right, each with up to MAX_BAND_TRY band widths -- taken from the record when this
				// is the seed k_ext4 extended ahead of the loop (the first one the loop reaches in a chain's main list)
				RgXExt xe = W.xe[ci - xc_lo];
				if (has_ext == 2 && pass == 0) { // a slot per seed, where the extension kernels left it: one 48-byte read, six lanes
					const unsigned long long *xp = XEw + (size_t)(l0 + si) * (sizeof(RgXExt) / 8);
					const unsigned long long v = lane < (int)(sizeof(RgXExt) / 8) ? xp[lane] : 0ull;
					unsigned long long w6[sizeof(RgXExt) / 8];
#pragma unroll
					for (int q = 0; q < (int)(sizeof(RgXExt) / 8); ++q) w6[q] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), q) << 32 | (unsigned)__builtin_amdgcn_readlane((int)v, q);
					memcpy(&xe, w6, sizeof(xe));
				}
				const bool cached = has_ext && pass == 0 && uni(xe.status) == 1 && uni(xe.si) == si;
				if (cached) ++pf_cached; else ++pf_inl;
				if (!cached && win_ok == 0) { // the chain's reference window (bns_fetch_seq, memchain.c:889) into LDS, once, when a seed of it is first extended
					const int span = (int)(rmax1 - rmax0);
					if (span > 0 && span <= WT::WINCAP) {
						dev_fetch_window(W.win, ix.pac, l_pac, rmax0, span, lane);
						win_ok = 1;
						WAVE_SYNC();
					} else win_ok = -1;
				}
				bsx_region_t R; memset(&R, 0, sizeof(R));
				int aw0 = P.w, aw1 = P.w;
				const int qe = s_qbeg + s_len;
				R.score = R.truesc = -1; R.rid = rid;
				if (cached) {
					R.rb = uni64(xe.rb); R.re = uni64(xe.re); R.qb = uni(xe.qb); R.qe = uni(xe.qe); R.score = uni(xe.score); R.truesc = uni(xe.truesc);
					aw0 = uni(xe.aw0); aw1 = uni(xe.aw1);
				} else
				for (int side = 0; side < 2; ++side) {
					if (side == 0 && s_qbeg == 0) { R.score = R.truesc = s_len * P.a; R.qb = 0; R.rb = s_rbeg; continue; }
					if (side == 1 && qe == l_query) { R.qe = l_query; R.re = s_rbeg + s_len; continue; }
					const int sc0 = R.score, clip = side ? P.pen_clip3 : P.pen_clip5;
					int aw = P.w;
					bsx_ext_res_t res; res.score = -1; res.qle = res.tle = res.gtle = 0; res.gscore = -1; res.max_off = 0;
					bsx_ext_job_t J;
					J.parent = (uint8_t)parent; J.pad = 0; J.end_bonus = clip;
					if (side == 0) { J.qoff = qoff + (uint32_t)s_qbeg - 1; J.qdir = -1; J.qlen = s_qbeg; J.tpos = s_rbeg - 1; J.tdir = -1; J.tlen = (int)(s_rbeg - rmax0); J.h0 = s_len * P.a; }
					else { J.qoff = qoff + (uint32_t)qe; J.qdir = 1; J.qlen = l_query - qe; J.tpos = s_rbeg + s_len; J.tdir = 1; J.tlen = (int)(rmax1 - (s_rbeg + s_len)); J.h0 = sc0; }
					for (int i = 0; i < 2; ++i) {
						const int prev = R.score;
						aw = P.w << i;
						J.w = aw;
						if (P.prof) { const long long now_ = (long long)__builtin_readcyclecounter(); RG_PF_ADD(W, 6, now_ - pf_t); pf_t = now_; }
						// rows in registers, 64 entries per lane slot: as few slots as the query needs (a row's cost grows with them)
						if (J.qlen < 64) res = ext_dp_reg<1>(ix, sc, reads, J, lane, win_ok > 0 ? W.win : nullptr, rmax0, W.q, qoff);
						else if (J.qlen < 128) res = ext_dp_reg<2>(ix, sc, reads, J, lane, win_ok > 0 ? W.win : nullptr, rmax0, W.q, qoff);
						else if (WT::QCAP <= 256 || J.qlen < 256) res = ext_dp_reg<RG_NC>(ix, sc, reads, J, lane, win_ok > 0 ? W.win : nullptr, rmax0, W.q, qoff);
						else if (P.ext_win && 2 * J.w + 1 <= 256 && (long long)J.h0 + (long long)J.qlen * (parent ? sc.mx_ct : sc.mx_ga) < (1 << 21))
							res = ext_dp_win<5>(ix, sc, reads, J, lane, win_ok > 0 ? W.win : nullptr, rmax0, W.q, qoff);   // rows in five register slots that follow the band (round 6)
						else if (WT::ROWS && 2 * J.w + 1 <= 512) { WAVE_SYNC(); res = ext_dp<8>(ix, sc, reads, J, W.Hrow, W.Erow, W.qrow, lane); }   // rows in LDS, the band (<= 8 x 64 columns) in registers
						else return 2;   // (-w above 127 with reads beyond 256 bases: left to the caller's batch kernels)
						res.score = uni(res.score); res.qle = uni(res.qle); res.tle = uni(res.tle); res.gtle = uni(res.gtle);
						res.gscore = uni(res.gscore); res.max_off = uni(res.max_off);
						R.score = res.score;
						if (P.prof) { const long long now_ = (long long)__builtin_readcyclecounter(); RG_PF_ADD(W, 7, now_ - pf_t); pf_t = now_; }
						++pf_ext; pf_rows += (unsigned int)(res.tle > res.gtle ? res.tle : res.gtle);
						if (R.score == prev || res.max_off < (aw >> 1) + (aw >> 2)) break;
					}
					const int local = res.gscore <= 0 || res.gscore <= R.score - clip;
					if (side == 0) {
						aw0 = aw;
						if (local) { R.qb = s_qbeg - res.qle; R.rb = s_rbeg - res.tle; R.truesc = R.score; }
						else { R.qb = 0; R.rb = s_rbeg - res.gtle; R.truesc = res.gscore; }
					} else {
						aw1 = aw;
						if (local) { R.qe = qe + res.qle; R.re = s_rbeg + s_len + res.tle; R.truesc += R.score - sc0; }
						else { R.qe = l_query; R.re = s_rbeg + s_len + res.gtle; R.truesc += res.gscore - sc0; }
					}
				}
				R.bss = (uint8_t)RG_BSS(parent, l_pac, R.rb); R.parent = (uint8_t)parent;
				if (RG_BSS(parent, l_pac, R.re) != R.bss) continue;   // crosses the strand boundary (memchain.c:846-849)
				int cov = 0;
				for (int i = lane; i < nl; i += 64) {
					const RgXSeed td = Lsd[i];
					if (td.qbeg >= R.qb && td.qbeg + td.len <= R.qe && td.rbeg >= R.rb && td.rbeg + td.len <= R.re) cov += td.len;
				}
				R.seedcov = uni(wave_sum_i32(cov));
				R.w = aw0 > aw1 ? aw0 : aw1; R.seedlen0 = s_len; R.frac_rep = frac_rep;
				if (uni(W.n_regs) == WT::XREGS) return 6;
				WAVE_SYNC();
				if (lane == 0) { W.regs[W.n_regs] = R; ++W.n_regs; }
				WAVE_SYNC();
			}
		}
	}
	if (P.prof) { RG_PF_ADD(W, 6, (long long)__builtin_readcyclecounter() - pf_t); RG_PF_ADD(W, 8, pf_ext); RG_PF_ADD(W, 9, pf_rows);
	              RG_PF_ADD(W, 11, pf_seen); RG_PF_ADD(W, 12, pf_skip); RG_PF_ADD(W, 13, pf_cached); RG_PF_ADD(W, 14, pf_inl); }
	return 0;
}

// ---- C3: mem_flt_chained_seeds (memchain.c:537-568) over the exported chains of the strand searches it applies to (reads long enough,
// or a small -W): every seed of a chain's main list shorter than MEM_SHORT_LEN is scored by ksw_align2 (16-bit, score only: the
// reference asks for the start too and never looks at it) over the seed +- MEM_SHORT_EXT on both sequences (mem_seed_sw,
// memchain.c:501-535); seeds below the threshold leave the list, the others carry their score into mem_chain2region1's best-first
// order.  One wavefront per strand search, the lists compacted in place ahead of k_c2r.
#include "sw_pass.hpp"
#define RG_SHORT_EXT 50
#define RG_SHORT_LEN 200
// Round 5: the alignments of a chunk's seeds are one batch.  A kilobase read has a few dozen seeds to score on its own strand (and the other
// strand's chance seeds as well), 16 M jobs of <= 199 x 199 cells per chunk; a wavefront per strand search with a wavefront per
// alignment (the form of rounds 3-4: 730 ms per chunk) spends two wave-wide prefix scans per row.  Now:
//   k_seedsw_prep   a wavefront per strand search, a lane per seed: the window of mem_seed_sw, the seed's job appended to the chunk's list;
//                   the seed remembers its job's number (bit 31 of its score field) until the scores are in
//   k_swl16         ksw_i16 (ksw.c:232-334) as Farrar's striped kernel itself, the 8 word lanes of its SSE register being 8 lanes of the
//                   wavefront: EIGHT jobs to a wave (k_swl.hip does the same for ksw_u8 with 16 lanes); score only
//   k_seedsw_apply  a wavefront per strand search: seeds below the threshold leave their lists, the lists are compacted in place
struct SswJob { long long rb; uint32_t qoff; short qlen, tlen; };   // qoff: bit 31 = the strand's matrix (parent)
#define SSW_PENDING 0x80000000u
#define SSW_BLK 1024u
#define SSW_NOROOM 0x3fffffffu   // the job list was full: k_seedsw_apply aligns the seed itself

// the window of mem_seed_sw (memchain.c:501-535) for a seed: false = no alignment is made (the seed keeps len * a)
__device__ __forceinline__ bool ssw_window(const DevIndex &ix, int l_query, long long s_rbeg, int s_qbeg, int s_len, int &qb_, int &qlen_, long long &rb_, int &tlen_)
{
	const long long l_pac = ix.l_pac;
	if (s_len >= RG_SHORT_LEN) return false;
	int qb = s_qbeg, qe = s_qbeg + s_len;
	long long rb = s_rbeg, re = s_rbeg + s_len;
	const long long mid = (rb + re) >> 1;
	qb -= RG_SHORT_EXT; qb = qb > 0 ? qb : 0;
	qe += RG_SHORT_EXT; qe = qe < l_query ? qe : l_query;
	rb -= RG_SHORT_EXT; rb = rb > 0 ? rb : 0;
	re += RG_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
	if (qe - qb >= RG_SHORT_LEN || re - rb >= RG_SHORT_LEN) return false;
	// bns_fetch_seq (bntseq.c:415-437): the span is cut to the contig of its middle
	const int is_rev = mid >= l_pac;
	const int rid = rg_pos2rid(ix, (const long long*)ix.ctg_off, rg_depos(l_pac, mid));
	long long far_beg = ix.ctg_off[rid], far_end = ix.ctg_off[rid + 1];
	if (is_rev) { const long long tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
	rb = rb > far_beg ? rb : far_beg; re = re < far_end ? re : far_end;
	qb_ = qb; qlen_ = qe - qb; rb_ = rb; tlen_ = (int)(re - rb);
	return qlen_ > 0 && tlen_ > 0;
}

// A kilobase read's strand search exports hundreds of chains of one or two seeds (chance 19-mers) and a few long ones: chains of up to
// SSW_SHORT seeds get a lane each, 64 chains at a time; the long ones a wavefront, 64 seeds at a time.
#define SSW_SHORT 8
struct SswAlloc { unsigned int at, left; };   // the wave's block of job numbers (one atomic on the chunk's counter per SSW_BLK jobs)
__device__ __forceinline__ unsigned int ssw_take(SswAlloc &A, bool want, SswJob *jobs, unsigned int job_cap, unsigned int *job_count, int lane)
{   // a job number for every lane that wants one (wave-uniform control flow); what a block leaves is filled with empty jobs
	const unsigned long long m = __ballot(want);
	if (m == 0) return 0;
	const unsigned int cnt = (unsigned int)__popcll(m);
	if (cnt > A.left) {
		SswJob Z; Z.rb = 0; Z.qoff = 0; Z.qlen = 0; Z.tlen = 0;
		for (unsigned int z = lane; z < A.left; z += 64) if (A.at + z < job_cap) jobs[A.at + z] = Z;
		unsigned int at = 0;
		if (lane == 0) at = atomicAdd(job_count, (unsigned int)SSW_BLK);
		A.at = (unsigned int)uni((int)__shfl((int)at, 0)); A.left = SSW_BLK;
	}
	const unsigned int id = A.at + (unsigned int)__popcll(m & ((1ull << lane) - 1));
	A.at += cnt; A.left -= cnt;
	return id;
}
__device__ __forceinline__ void ssw_prep_seed(const DevIndex &ix, SswAlloc &A, bool have, RgXSeed *slot, int l_query, uint32_t qoff, int parent,
                                              SswJob *jobs, unsigned int job_cap, unsigned int *job_count, int lane)
{
	bool want = false;
	int qb = 0, qlen = 0, tlen = 0; long long rb = 0;
	RgXSeed sd; sd.rbeg = 0; sd.qbeg = sd.len = 0; sd.sb = 0;
	if (have) { sd = *slot; want = ssw_window(ix, l_query, sd.rbeg, sd.qbeg, sd.len, qb, qlen, rb, tlen); }
	const unsigned int id = ssw_take(A, want, jobs, job_cap, job_count, lane);
	if (want) {
		if (id < job_cap) {
			SswJob J; J.rb = rb; J.qoff = (qoff + (uint32_t)qb) | (parent ? 0x80000000u : 0u); J.qlen = (short)qlen; J.tlen = (short)tlen;
			jobs[id] = J;
			slot->sb = (int)(SSW_PENDING | id << 1 | (unsigned int)XS_BAD(sd));
		} else slot->sb = (int)(SSW_PENDING | SSW_NOROOM << 1 | (unsigned int)XS_BAD(sd));
	}
}

__global__ void __launch_bounds__(64, 8)
k_seedsw_prep(DevIndex ix, const bsx_seed_task_t *tasks, RgXPool X, unsigned int *cursor, SswJob *jobs, unsigned int job_cap, unsigned int *job_count)
{
	const int lane = wave_lane();
	const int n = (int)*X.xcount;
	SswAlloc A; A.at = 0; A.left = 0;
	for (;;) {
		int i = 0;
		if (lane == 0) i = (int)atomicAdd(cursor, 1u);
		i = uni(__shfl(i, 0));
		if (i >= n) break;
		const int t = uni(X.xlist[i]);
		RgXHdr *H = (RgXHdr*)(X.base + uni64(X.xoff[t]));
		if (uni(H->flt) == RG_NOFLT) continue;
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const int nk = uni(H->n_chains);
		RgXChain *XC = (RgXChain*)(H + 1);
		RgXSeed *XS = (RgXSeed*)(XC + nk);
		for (int cbase = 0; cbase < nk; cbase += 64) {
			const int c = cbase + lane;
			int seed_off = 0, n_main = 0;
			if (c < nk) { seed_off = XC[c].seed_off; n_main = (int)XC[c].n_main; }
			const bool shortc = n_main <= SSW_SHORT;
			const int rounds = wave_max_i32(shortc ? n_main : 0);
			for (int j = 0; j < rounds; ++j)
				ssw_prep_seed(ix, A, shortc && j < n_main, XS + seed_off + j, l_query, qoff, parent, jobs, job_cap, job_count, lane);
			unsigned long long longs = __ballot(!shortc);
			while (longs) {
				const int src = __ffsll((long long)longs) - 1;
				longs &= longs - 1;
				const int so = uni(__shfl(seed_off, src)), nm = uni(__shfl(n_main, src));
				for (int base = 0; base < nm; base += 64)
					ssw_prep_seed(ix, A, base + lane < nm, XS + so + base + lane, l_query, qoff, parent, jobs, job_cap, job_count, lane);
			}
		}
	}
	SswJob Z; Z.rb = 0; Z.qoff = 0; Z.qlen = 0; Z.tlen = 0;
	for (unsigned int z = lane; z < A.left; z += 64) if (A.at + z < job_cap) jobs[A.at + z] = Z;
}

// ---- ksw_i16 (ksw.c:232-334), score only, SIXTEEN jobs to a wavefront.  A job is 8 lanes -- the 8 words of the SSE register (k_swl.hip does
// the same for ksw_u8 with 16 lanes) -- and every lane carries two jobs, one in each half of its 32-bit registers (v_pk_add_u16 / v_pk_sub_u16
// clamp / v_pk_max_i16: the SSE instructions themselves, two jobs at a time).
//   * all 16 jobs run with the stripe count of the longest query among them.  The local-alignment matrix does not depend on how its columns are
//     dealt to stripes, and the padding columns behind a query (profile 0, as ksw_qinit pads its last stripe) cannot raise the maximum: what
//     reaches them came from a cell that was counted already.
//   * the query profile is kept biased (score + bias as an unsigned byte, four target bases to a register); a row's scores for both jobs come
//     from one v_perm_b32.  h = max(0, h + s) instead of h + s: the next instruction is a maximum with E and F, which are >= 0.
//   * E and F floor at 0 as in ksw_i16 (_mm_subs_epu16); the scores fit 15 bits (199 columns x an int8 match score)
//   * the lazy-F loop (ksw.c:285-295) is what it converges to: F carried in from the job's lower lanes by a prefix maximum over the lanes'
//     last F values (k_swl.hip, swl_row, has the argument)
#define SSW_SL 25   // stripes: ceil(199 / 8)
typedef unsigned short ssw_u2 __attribute__((ext_vector_type(2)));
typedef short ssw_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ssw_u2, a) + __builtin_bit_cast(ssw_u2, b)); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ssw_u2, a), __builtin_bit_cast(ssw_u2, b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ssw_s2, a), __builtin_bit_cast(ssw_s2, b))); }
__device__ __forceinline__ uint32_t pk_dup(int v) { return (uint32_t)(v & 0xffff) * 0x10001u; }
__device__ __forceinline__ uint32_t ssw8_or(uint32_t v)   // OR over the job's 8 lanes
{
	v |= (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
	v |= (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
	v |= (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false);   // row_half_mirror: the other quad of the 8
	return v;
}
__device__ __forceinline__ uint32_t ssw8_pkmax(uint32_t v)
{
	v = pk_max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));
	v = pk_max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));
	v = pk_max(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));
	return v;
}
// the lane's column of every stripe against target bases 0..3, biased.  lut: the register for each read base (0..4, then the padding column's),
// the two matrices one behind the other
__device__ __forceinline__ void ssw_profile(uint32_t (&prof)[SSW_SL], const uint8_t *reads, const uint32_t *lut, uint32_t qoff, int qlen, int slen, int k)
{
	int q[SSW_SL];
#pragma unroll
	for (int j = 0; j < SSW_SL; ++j) {
		const int c = j + k * slen;   // word k of stripe j is query column j + k * slen (ksw.c:100-107)
		q[j] = reads[(size_t)qoff + (c < qlen ? c : 0)];   // (all the loads in flight together)
		q[j] = (j < slen && c < qlen && q[j] < 5) ? q[j] : 5;
	}
#pragma unroll
	for (int j = 0; j < SSW_SL; ++j) prof[j] = lut[q[j]];
}
template<bool UNI>   // UNI: deletions and insertions cost the same
__global__ void __launch_bounds__(256)
k_swl16(DevIndex ix, DevScoring sc, const uint8_t *reads, const SswJob *jobs, const unsigned int *job_count, unsigned int job_cap, int *scores)
{
	const int lane = wave_lane(), g = lane >> 3, k = lane & 7;
	unsigned int n = *job_count; if (n > job_cap) n = job_cap;
	const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (blockDim.x >> 6);
	int bias = 0;
	for (int i = 0; i < 25; ++i) { bias = (int)sc.ctmat[i] < -bias ? -(int)sc.ctmat[i] : bias; bias = (int)sc.gamat[i] < -bias ? -(int)sc.gamat[i] : bias; }
	__shared__ uint32_t lut[16];   // [0..5] gamat (parent 0), [8..13] ctmat
	if (threadIdx.x < 16) {
		const int q = threadIdx.x & 7;
		const int8_t *mat = threadIdx.x >> 3 ? sc.ctmat : sc.gamat;
		lut[threadIdx.x] = q > 4 ? (uint32_t)bias * 0x01010101u
		                         : ((uint32_t)(uint8_t)(mat[q] + bias) | (uint32_t)(uint8_t)(mat[5 + q] + bias) << 8 | (uint32_t)(uint8_t)(mat[10 + q] + bias) << 16 | (uint32_t)(uint8_t)(mat[15 + q] + bias) << 24);
	}
	__syncthreads();
	const uint32_t biasv = pk_dup(bias), oe_delv = pk_dup(sc.o_del + sc.e_del), oe_insv = pk_dup(sc.o_ins + sc.e_ins), e_delv = pk_dup(sc.e_del), e_insv = pk_dup(sc.e_ins);
	for (long long q16 = wave_id; q16 * 16 < (long long)n; q16 += n_waves) {
		const long long jA = q16 * 16 + (g << 1), jB = jA + 1;
		const bool vA = jA < (long long)n, vB = jB < (long long)n;
		const SswJob JA = jobs[vA ? jA : 0], JB = jobs[vB ? jB : 0];
		const int qlenA = vA ? (int)JA.qlen : 0, tlenA = vA ? (int)JA.tlen : 0, qlenB = vB ? (int)JB.qlen : 0, tlenB = vB ? (int)JB.tlen : 0;
		const int slen = uni(wave_max_i32(((qlenA > qlenB ? qlenA : qlenB) + 7) >> 3));
		const int rows = uni(wave_max_i32(tlenA > tlenB ? tlenA : tlenB));
		if (slen == 0 || slen > SSW_SL) { if (k == 0) { if (vA) scores[jA] = 0; if (vB) scores[jB] = 0; } continue; }   // (empty jobs: what a wave left of its block)
		uint32_t H[SSW_SL], E[SSW_SL], profA[SSW_SL], profB[SSW_SL];
		ssw_profile(profA, reads, lut + ((JA.qoff >> 31) << 3), JA.qoff & 0x7fffffffu, qlenA, slen, k);
		ssw_profile(profB, reads, lut + ((JB.qoff >> 31) << 3), JB.qoff & 0x7fffffffu, qlenB, slen, k);
#pragma unroll
		for (int j = 0; j < SSW_SL; ++j) H[j] = E[j] = 0;
		const uint32_t col0ev = pk_dup(k * slen * sc.e_ins), col1ev = pk_dup(k ? (k - 1) * slen * sc.e_ins : 0);
		uint32_t gmaxv = 0, hlast = 0;
		unsigned int twA = 0, twB = 0;   // the target bases of 8 rows, two bits each, the same in the job's 8 lanes
		for (int i = 0; i < rows; ++i) {
			if ((i & 7) == 0) {
				const int ii = i + k;
				twA = ssw8_or(ii < tlenA ? (unsigned int)dev_ref_base(ix.pac, ix.l_pac, JA.rb + ii) << (k << 1) : 0u);
				twB = ssw8_or(ii < tlenB ? (unsigned int)dev_ref_base(ix.pac, ix.l_pac, JB.rb + ii) << (k << 1) : 0u);
			}
			const int sh = (i & 7) << 1;
			const uint32_t sel = ((twA >> sh) & 3u) | 0x0c000c00u | (4u + ((twB >> sh) & 3u)) << 16;   // byte 0: job A's score, byte 2: job B's, bytes 1 and 3: zero
			const uint32_t livem = (i < tlenA ? 0xffffu : 0u) | (i < tlenB ? 0xffff0000u : 0u);
			// the striped main loop (ksw.c:268-284)
			uint32_t f = 0, mxv = 0;
			const int hs = __builtin_amdgcn_update_dpp(0, (int)hlast, DPP_ROW_SHR(1), 0xf, 0xf, false);
			uint32_t h = k ? (uint32_t)hs : 0u;   // _mm_slli_si128(H[slen - 1], 2)
#pragma unroll
			for (int j = 0; j < SSW_SL; ++j) {
				if (j >= slen) { hlast = H[j > 0 ? j - 1 : 0]; break; }
				h = pk_subs(pk_add(h, __builtin_amdgcn_perm(profB[j], profA[j], sel)), biasv);
				uint32_t e = E[j];
				h = pk_max(h, e);
				h = pk_max(h, f);
				mxv = pk_max(mxv, h);
				const uint32_t hold = H[j];
				H[j] = h;
				const uint32_t t = pk_subs(h, oe_delv);
				E[j] = pk_max(pk_subs(e, e_delv), t);
				const uint32_t t2 = UNI ? t : pk_subs(h, oe_insv);
				f = pk_max(pk_subs(f, e_insv), t2);
				h = hold;
			}
			if (slen == SSW_SL) hlast = H[SSW_SL - 1];
			// f is now F of the column behind the lane's last one -- the next lane's first (what `f = _mm_slli_si128(f, 2)` hands on); through a
			// whole lane it decays by slen * e_ins: lane k's first column gets max over k' < k of f(k') - (k - 1 - k') * slen * e_ins
			uint32_t gl = pk_add(f, col0ev);
			uint32_t t;
			t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gl, DPP_ROW_SHR(1), 0xf, 0xf, false); gl = pk_max(gl, k >= 1 ? t : 0u);
			t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gl, DPP_ROW_SHR(2), 0xf, 0xf, false); gl = pk_max(gl, k >= 2 ? t : 0u);
			t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gl, DPP_ROW_SHR(4), 0xf, 0xf, false); gl = pk_max(gl, k >= 4 ? t : 0u);
			t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gl, DPP_ROW_SHR(1), 0xf, 0xf, false);
			uint32_t v = pk_subs(k >= 1 ? t : 0u, col1ev);
			if (__ballot(v != 0) != 0) {
#pragma unroll
				for (int j = 0; j < SSW_SL; ++j) {
					if (j >= slen) { hlast = H[j > 0 ? j - 1 : 0]; break; }
					H[j] = pk_max(H[j], v); v = pk_subs(v, e_insv);
				}
				if (slen == SSW_SL) hlast = H[SSW_SL - 1];
			}
			gmaxv = pk_max(gmaxv, mxv & livem);
		}
		gmaxv = ssw8_pkmax(gmaxv);
		if (k == 0) { if (vA) scores[jA] = (int)(gmaxv & 0xffffu); if (vB) scores[jB] = (int)(gmaxv >> 16); }
	}
}

// one seed's score for k_seedsw_apply: from the batch, or -- no room in the chunk's job list -- aligned here, a wavefront for the one
// alignment (the form of rounds 3-4).  Wave-uniform control flow; `have`: this lane holds a seed.
__device__ __forceinline__ int ssw_score_of(const DevIndex &ix, const DevScoring &sc, const int8_t *mat, const uint8_t *reads, uint32_t qoff, int l_query,
                                            bool have, const RgXSeed &sd, const int *scores, unsigned int &n_sw, int lane)
{
	const unsigned int sbu = (unsigned int)sd.sb;
	const bool pending = have && (sbu & SSW_PENDING);
	const unsigned int id = (sbu >> 1) & 0x3fffffffu;
	int score = -1;
	if (pending && id != SSW_NOROOM) score = scores[id];
	unsigned long long todo = __ballot(pending && id == SSW_NOROOM);
	while (todo) {
		const int src = __ffsll((long long)todo) - 1;
		todo &= todo - 1;
		const long long s_rbeg = uni64(__shfl((long long)sd.rbeg, src)); const int s_qbeg = uni(__shfl((int)sd.qbeg, src)), s_len1 = uni(__shfl((int)sd.len, src));
		int qb, qlen, tlen; long long rb; int sc1 = -1;
		if (ssw_window(ix, l_query, s_rbeg, s_qbeg, s_len1, qb, qlen, rb, tlen)) {
			const int Q = ((qlen + 7) >> 3) << 3;
#define SEEDSW_RUN(NC_) do { int qv[NC_]; _Pragma("unroll") for (int c = 0; c < NC_; ++c) { const int jj = (c << 6) + lane; qv[c] = jj < qlen ? (int)reads[(size_t)qoff + qb + jj] : 5; } \
				const SwPass r = sw_pass<NC_>(ix, mat, 0, qlen, qv, tlen, rb, 1, -1, sc.o_del, sc.e_del, sc.o_ins, sc.e_ins, 0, nullptr, lane); sc1 = uni(r.score); } while (0)
			if (Q <= 128) SEEDSW_RUN(2); else if (Q <= 192) SEEDSW_RUN(3); else SEEDSW_RUN(4);
		}
		if (lane == src) score = sc1;
	}
	n_sw += (unsigned int)__popcll(__ballot(pending));
	return score;
}

__global__ void __launch_bounds__(64, 4)
k_seedsw_apply(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks, RgXPool X, unsigned int *cursor, const int *scores, unsigned long long *counters)
{
	const int lane = wave_lane();
	const int n = (int)*X.xcount;
	unsigned int n_sw = 0;
	for (;;) {
		int i = 0;
		if (lane == 0) i = (int)atomicAdd(cursor, 1u);
		i = uni(__shfl(i, 0));
		if (i >= n) break;
		const int t = uni(X.xlist[i]);
		RgXHdr *H = (RgXHdr*)(X.base + uni64(X.xoff[t]));
		const int flt = uni(H->flt);
		if (flt == RG_NOFLT) continue;
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const int8_t *mat = parent ? sc.ctmat : sc.gamat;
		const int nk = uni(H->n_chains);
		RgXChain *XC = (RgXChain*)(H + 1);
		RgXSeed *XS = (RgXSeed*)(XC + nk);
		for (int cbase = 0; cbase < nk; cbase += 64) {
			const int c = cbase + lane;
			int seed_off = 0, n_main = 0, n_extra = 0;
			if (c < nk) { seed_off = XC[c].seed_off; n_main = (int)XC[c].n_main; n_extra = (int)XC[c].n_extra; }
			const bool shortc = n_main <= SSW_SHORT;
			// short chains: a lane each, its seeds one after the other (nobody else touches the chain's lists)
			const int rounds = wave_max_i32(shortc ? n_main : 0);
			int k = 0;
			for (int j = 0; j < rounds; ++j) {
				const bool have = shortc && j < n_main;
				RgXSeed sd; sd.rbeg = 0; sd.qbeg = sd.len = 0; sd.sb = 0;
				if (have) sd = XS[seed_off + j];
				const int score = ssw_score_of(ix, sc, mat, reads, qoff, l_query, have, sd, scores, n_sw, lane);
				if (have && (score < 0 || score >= flt)) {
					sd.sb = (score < 0 ? (int)sd.len * P.a : score) << 1 | (int)((unsigned int)sd.sb & 1u);
					XS[seed_off + k] = sd;   // k <= j
					++k;
				}
			}
			if (shortc && c < nk && k != n_main) {
				for (int e = 0; e < n_extra; ++e) XS[seed_off + k + e] = XS[seed_off + n_main + e];   // the backup list moves up behind the shortened main list
				XC[c].n_main = (unsigned short)k;
			}
			// long chains: a wavefront each, 64 seeds at a time -- what stays moves down behind what stayed before
			unsigned long long longs = __ballot(!shortc);
			while (longs) {
				const int src = __ffsll((long long)longs) - 1;
				longs &= longs - 1;
				const int so = uni(__shfl(seed_off, src)), nm = uni(__shfl(n_main, src)), ne = uni(__shfl(n_extra, src));
				int kk = 0;
				for (int base = 0; base < nm; base += 64) {
					const int j = base + lane;
					RgXSeed sd; sd.rbeg = 0; sd.qbeg = sd.len = 0; sd.sb = 0;
					if (j < nm) sd = XS[so + j];
					const int score = ssw_score_of(ix, sc, mat, reads, qoff, l_query, j < nm, sd, scores, n_sw, lane);
					const bool keep = j < nm && (score < 0 || score >= flt);
					const unsigned long long km = __ballot(keep);
					sd.sb = (score < 0 ? (int)sd.len * P.a : score) << 1 | (int)((unsigned int)sd.sb & 1u);
					WAVE_SYNC();   // every lane has read its seed before any is written (kk + rank <= j)
					if (keep) XS[so + kk + __popcll(km & ((1ull << lane) - 1))] = sd;
					WAVE_SYNC();
					kk += __popcll(km);
				}
				if (kk != nm) {
					const int shift = nm - kk;
					for (int base = 0; base < ne; base += 64) { // (targets never pass the sources of a later round)
						const int e = base + lane;
						RgXSeed v; v.rbeg = 0; v.qbeg = v.len = 0; v.sb = 0;
						if (e < ne) v = XS[so + nm + e];
						WAVE_SYNC();
						if (e < ne) XS[so + nm + e - shift] = v;
						WAVE_SYNC();
					}
					if (lane == 0) XC[src + cbase].n_main = (unsigned short)kk;
				}
			}
		}
	}
	if (P.prof && lane == 0 && n_sw) atomicAdd(&counters[42], (unsigned long long)n_sw);
}

#ifndef C2R_WPB
#define C2R_WPB 1    // waves per workgroup (a workgroup's slots come back when its last wave ends)
#endif
template <typename WT, int OCC>
__global__ void __launch_bounds__(64 * C2R_WPB, OCC)
k_c2r(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks, RgXPool X,
      bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
      unsigned int *cursor, int *next_list, unsigned int *next_count, unsigned long long *counters, int quota, unsigned char *slab)
{
	__shared__ WT lds[C2R_WPB];
	__shared__ int gap_tab[WT::GAPCAP + 1];
	__shared__ long long ctg_lds[RG_CTG_LDS + 1];
	P.gap_cap = WT::GAPCAP;
	for (int q = threadIdx.x; q <= WT::GAPCAP; q += blockDim.x) gap_tab[q] = rg_cal_max_gap(P, q);
	if (ix.n_seqs <= RG_CTG_LDS) for (int q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) ctg_lds[q] = ix.ctg_off[q];
	const long long *ctg_tab = ix.n_seqs <= RG_CTG_LDS ? (const long long*)ctg_lds : (const long long*)ix.ctg_off;
	__syncthreads();
	const int lane = wave_lane();
	WT &W = lds[threadIdx.x >> 6];
	RG_PF_ZERO(W);
	if constexpr (WT::HBM) { // the large tables: this wave's slab
		unsigned char *sl = slab + ((size_t)blockIdx.x * C2R_WPB + (threadIdx.x >> 6)) * WT::slab_bytes();
		if (lane == 0) W.regs = (bsx_region_t*)sl;
		WAVE_SYNC();
	}
	const int n = (int)*X.xcount;
	// a wave takes `quota` strand searches and leaves (the launch covers the worst case): workgroups with a bounded life let the
	// back half's short high-priority batches (k_sw, k_global) of an older chunk get compute units while this one runs
	for (int taken = 0; taken < quota; ++taken) {
		int i = 0;
		if (lane == 0) i = (int)atomicAdd(cursor, 1u);
		i = uni(__shfl(i, 0));
		if (i >= n) break;
		const int t = uni(X.xlist[i]);
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const RgXHdr *H = (const RgXHdr*)(X.base + uni64(X.xoff[t]));
		int status = rg_c2r(W, ix, sc, P, reads, l_query, parent, qoff, H, lane, counters, gap_tab, ctg_tab);
		WAVE_SYNC();
		if constexpr (WT::HBM) { // publish: hundreds of regions, all lanes copy
			const int nr = status ? 0 : uni(W.n_regs);
			unsigned long long base = 0;
			if (nr > 0) {
				if (lane == 0) base = atomicAdd(out_cursor, (unsigned long long)nr);
				base = (unsigned long long)uni64((long long)base);
				if (base + nr <= out_cap) { const bsx_region_t *src = W.regs; for (int k = lane; k < nr; k += 64) out[base + k] = src[k]; }
				else status = 7;
			}
			if (lane == 0) {
				reg_off[t] = (long long)base;
				reg_n[t] = status ? -status : nr;
				if (next_list && (status == 2 || status == 6)) next_list[atomicAdd(next_count, 1u)] = t;
			}
		} else
		if (lane == 0) { // publish (as rg_publish)
			const int nr = status ? 0 : W.n_regs;
			unsigned long long base = 0;
			if (nr > 0) {
				base = atomicAdd(out_cursor, (unsigned long long)nr);
				if (base + nr <= out_cap) for (int k = 0; k < nr; ++k) out[base + k] = W.regs[k];
				else status = 7;
			}
			reg_off[t] = (long long)base;
			reg_n[t] = status ? -status : nr;
			if (next_list && (status == 2 || status == 6)) { next_list[atomicAdd(next_count, 1u)] = t; if (P.prof) atomicAdd(counters + 160 + status, 1ull); }
		}
		WAVE_SYNC();
	}
	RG_PF_FLUSH(W);
}

// Debug builds of the LDS tiers (tools/dbg/lds_variants.sh; never the product's): RG_DBG 1 = guard words around every LDS object of the
// workgroup, checked when it ends (tests/test_gpu_lds_guards.py); 2 = the wave's tables filled with 0xff before every strand search (an
// answer that changes with the fill reads LDS it did not write)
#ifndef RG_DBG
#define RG_DBG 0
#endif
#if RG_DBG == 2
#define RG_DBG_FILL(S, D) do { WAVE_SYNC(); for (int i_ = lane; i_ < (int)(sizeof(S) / 4); i_ += 64) ((int*)&(S))[i_] = -1; \
		for (int i_ = (int)(sizeof((D).pf) / 4) + lane; i_ < (int)(sizeof(D) / 4); i_ += 64) ((int*)&(D))[i_] = -1; WAVE_SYNC(); } while (0)
#else
#define RG_DBG_FILL(S, D) do { } while (0)
#endif
#if RG_DBG == 1
#define RG_GUARD_WORD(q) ((int)0xC0DE0000 + (q))
#define RG_DBG_GUARDS(name) do { __syncthreads(); for (int q_ = threadIdx.x; q_ < 64; q_ += blockDim.x) { const int w_ = RG_GUARD_WORD(q_); \
		if (DL.c0[q_] != w_ || DL.c1[q_] != w_ || DL.c2[q_] != w_ || DL.c3[q_] != w_) \
			printf("LDS GUARD %s block %d word %d: %08x %08x %08x %08x\n", name, (int)blockIdx.x, q_, DL.c0[q_], DL.c1[q_], DL.c2[q_], DL.c3[q_]); } } while (0)
#else
#define RG_DBG_GUARDS(name) do { } while (0)
#endif
// first tier: tables in LDS.  Tasks declined for table size (or for tied chain starts) go on retry_list for the second tier.
#ifndef RG_WPB
#define RG_WPB 1     // waves per workgroup of the LDS tiers, as C2R_WPB
#endif
template <int OCC, typename DPT>
__global__ void __launch_bounds__(64 * RG_WPB, OCC)
k_regions(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks,
          const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
          bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
          unsigned int *task_cursor, int *retry_list, unsigned int *retry_count, int quota, unsigned long long *counters,
          const long long *pos_off, const unsigned long long *pos, const unsigned char *cls, RgXPool X)
{
#if RG_DBG == 1
	struct DbgLds { int c0[64]; RgSmall lds[RG_WPB]; int c1[64]; DPT dp[RG_WPB]; int c2[64]; long long ctg_lds[RG_CTG_LDS + 1]; int c3[64]; };
	__shared__ DbgLds DL;
	RgSmall *lds = DL.lds; DPT *dp = DL.dp; long long *ctg_lds = DL.ctg_lds;
	for (int q = threadIdx.x; q < 64; q += blockDim.x) { DL.c0[q] = DL.c1[q] = DL.c2[q] = DL.c3[q] = RG_GUARD_WORD(q); }
#else
	__shared__ RgSmall lds[RG_WPB];
	__shared__ DPT dp[RG_WPB];
	__shared__ long long ctg_lds[RG_CTG_LDS + 1];
#endif
	const int *gap_tab = nullptr;   // (cal_max_gap is read by the chain-to-region loop, which this tier leaves to k_c2r: its table was a kilobyte of LDS per workgroup)
	P.gap_cap = -1;
	if (ix.n_seqs <= RG_CTG_LDS) for (int q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) ctg_lds[q] = ix.ctg_off[q];
	const long long *ctg_tab = ix.n_seqs <= RG_CTG_LDS ? (const long long*)ctg_lds : (const long long*)ix.ctg_off;
	__syncthreads();
	const int lane = wave_lane();
	RgSmall &S = lds[threadIdx.x >> 6];
	DPT &D = dp[threadIdx.x >> 6];
	RG_PF_ZERO(D);
	// each wave takes `quota` tasks and leaves (bounded workgroup life, see k_seed); the launch covers all tasks
	for (int taken = 0; taken < quota; ++taken) {
		int t = 0;
		if (lane == 0) t = (int)atomicAdd(task_cursor, 1u);
		t = uni(__shfl(t, 0));
		if (t >= n_tasks) break;
		const int cl = cls ? uni((int)cls[t]) : 0;
		if (cl != 0) { // larger than this tier's tables: straight to the next one (3: the last HBM tier has it already, launch_occ's early list)
			if (lane == 0 && cl != 3) retry_list[atomicAdd(retry_count, 1u)] = t;
			--taken;   // costs nothing: the quota counts strand searches done here
			continue;
		}
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent), n_iv = uni(task_n[t]);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const long long po = pos ? uni64(pos_off[t]) : -1;
		RG_DBG_FILL(S, D);
		int status = rg_task<RgSmall, true, DPT>(S, D, ix, sc, P, reads, l_query, parent, qoff, seeds_dense + uni64(task_off[t]), n_iv, po >= 0 ? pos + po : nullptr, lane, counters, gap_tab, ctg_tab, &X, t);
		if (status == 11) continue;   // exported: k_c2r makes and publishes its regions
		status = rg_publish(S, t, status, out, out_cap, out_cursor, reg_off, reg_n, lane);
		if ((status == 8 || status == 2 || status == 3 || status == 4 || status == 6 || status == 10) && lane == 0) retry_list[atomicAdd(retry_count, 1u)] = t;   // (10: the HBM tiers walk an over-represented interval further)
	}
	RG_PF_FLUSH(D);
	RG_DBG_GUARDS("k_regions");
}

// second and third tier: the same code over per-wave tables in HBM, for the strand searches of repeat-rich reads.
// `list` names the tasks (null: 0..*count-1); what this tier declines for table size goes on next_list.
template <typename Store, bool XSPLIT, typename DPT>
__global__ void __launch_bounds__(256, 3)
k_regions_slab(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks,
               const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
               bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
               const int *list, const unsigned int *count, unsigned int *cursor, Store *slabs, int *next_list, unsigned int *next_count,
               unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, RgXPool X)
{
	__shared__ DPT dp[4];
	__shared__ int gap_tab[DPT::QCAP + 1];
	__shared__ long long ctg_lds[RG_CTG_LDS + 1];
	P.gap_cap = DPT::QCAP;
	for (int q = threadIdx.x; q <= DPT::QCAP; q += blockDim.x) gap_tab[q] = rg_cal_max_gap(P, q);
	if (ix.n_seqs <= RG_CTG_LDS) for (int q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) ctg_lds[q] = ix.ctg_off[q];
	const long long *ctg_tab = ix.n_seqs <= RG_CTG_LDS ? (const long long*)ctg_lds : (const long long*)ix.ctg_off;
	__syncthreads();
	const int lane = wave_lane();
	Store &S = slabs[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)];
	DPT &D = dp[threadIdx.x >> 6];
	RG_PF_ZERO(D);
	unsigned long long wave_cyc = 0;
	const int n = (int)*count;
	for (;;) {
		int i = 0;
		if (lane == 0) i = (int)atomicAdd(cursor, 1u);
		i = uni(__shfl(i, 0));
		if (i >= n) break;
		const int t = list ? uni(list[i]) : i;
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent), n_iv = uni(task_n[t]);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const long long po = pos ? uni64(pos_off[t]) : -1;
		const long long tk0 = P.prof ? (long long)__builtin_readcyclecounter() : 0;
		// status 10: an over-represented interval has to be walked past its first max_occ occurrences; again with eight times as many of it, ...
		int bi[4], bl[4], nb = 0, status;
		for (;;) {
			status = rg_task<Store, XSPLIT, DPT>(S, D, ix, sc, P, reads, l_query, parent, qoff, seeds_dense + uni64(task_off[t]), n_iv, po >= 0 ? pos + po : nullptr, lane, counters, gap_tab, ctg_tab, &X, t, nb, bi, bl);
			if (status != 10 || !Store::NODES || !P.walk_on) break;
			WAVE_SYNC();
			const int iv = uni(S.boost_iv);
			int j = 0;
			while (j < nb && bi[j] != iv) ++j;
			if (j == nb) { if (nb == 4) break; bi[nb] = iv; bl[nb] = 8; ++nb; }
			else if (bl[j] >= (1 << 20)) break;
			else bl[j] <<= 3;
			WAVE_SYNC();
		}
		if (P.prof && lane == 0) { // $BSX_PHASES: the longest strand search of the launch, their sum and number, the busiest wave
			const unsigned long long dt = (unsigned long long)((long long)__builtin_readcyclecounter() - tk0);
			unsigned long long *c = counters + (Store::SCAP > 1024 ? 114 : 110);
			atomicMax(c, dt); atomicAdd(c + 1, dt); atomicAdd(c + 2, 1ull); wave_cyc += dt;
			if (Store::SCAP > 1024) { // the last tier: its strand searches by duration (powers of two of 2^16 cycles), and what the longest one looked like
				const unsigned long long b = dt >> 16;
				atomicAdd(counters + 130 + (b ? 64 - __builtin_clzll(b) : 0), 1ull);
				const unsigned long long nk_ = (unsigned long long)(S.n_chains < 0 ? 0 : S.n_chains > 16383 ? 16383 : S.n_chains), nr_ = (unsigned long long)(S.n_regs < 0 ? 0 : S.n_regs > 16383 ? 16383 : S.n_regs);
				atomicMax(counters + 128, (dt >> 14) << 38 | (unsigned long long)(n_iv > 4095 ? 4095 : n_iv) << 26 | (nk_ & 8191) << 13 | (nr_ & 8191));
			}
		}
		if (status == 11) continue;   // exported (XSPLIT: chunks with long reads, whose chains go through k_seedsw and k_c2r)
		status = rg_publish(S, t, status, out, out_cap, out_cursor, reg_off, reg_n, lane);
		if (next_list && (status == 8 || status == 2 || status == 3 || status == 4 || status == 6) && lane == 0) next_list[atomicAdd(next_count, 1u)] = t;
	}
	RG_PF_FLUSH(D);
	if (P.prof && lane == 0 && wave_cyc) atomicMax(counters + (Store::SCAP > 1024 ? 117 : 113), wave_cyc);
}

// Between the first tier and the HBM tiers: the same tables four times larger, still in LDS (two waves per workgroup, three
// workgroups per CU).  On a larger genome a repeat family has more copies, and a quarter of the strand searches outgrow the
// first tier's 96 seeds; in HBM slabs every table access is a memory round trip.  Same list protocol as k_regions_slab.
#define MID_WPB 2    // 14 KB of tables per wave: five workgroups of two waves fit a CU, only nine of one
template <typename Store, typename DPT, int WPB, int OCC>
__global__ void __launch_bounds__(64 * WPB, OCC)
k_regions_mid(DevIndex ix, DevScoring sc, RegParams P, const uint8_t *reads, const bsx_seed_task_t *tasks,
              const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
              bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
              const int *list, const unsigned int *count, unsigned int *cursor, int *next_list, unsigned int *next_count,
              unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, RgXPool X, int quota)
{
#if RG_DBG == 1
	struct DbgLds { int c0[64]; Store lds[WPB]; int c1[64]; DPT dp[WPB]; int c2[64]; long long ctg_lds[RG_CTG_LDS + 1]; int c3[64]; };
	__shared__ DbgLds DL;
	Store *lds = DL.lds; DPT *dp = DL.dp; long long *ctg_lds = DL.ctg_lds;
	for (int q = threadIdx.x; q < 64; q += blockDim.x) { DL.c0[q] = DL.c1[q] = DL.c2[q] = DL.c3[q] = RG_GUARD_WORD(q); }
#else
	__shared__ Store lds[WPB];
	__shared__ DPT dp[WPB];
	__shared__ long long ctg_lds[RG_CTG_LDS + 1];
#endif
	const int *gap_tab = nullptr;   // (as in k_regions: this tier stops before the loop that reads cal_max_gap)
	P.gap_cap = -1;
	if (ix.n_seqs <= RG_CTG_LDS) for (int q = threadIdx.x; q <= ix.n_seqs; q += blockDim.x) ctg_lds[q] = ix.ctg_off[q];
	const long long *ctg_tab = ix.n_seqs <= RG_CTG_LDS ? (const long long*)ctg_lds : (const long long*)ix.ctg_off;
	__syncthreads();
	const int lane = wave_lane();
	Store &S = lds[threadIdx.x >> 6];
	DPT &D = dp[threadIdx.x >> 6];
	RG_PF_ZERO(D);
	const int n = (int)*count;
	for (int taken = 0; taken < quota; ++taken) {   // bounded workgroup life, as in k_c2r
		int i = 0;
		if (lane == 0) i = (int)atomicAdd(cursor, 1u);
		i = uni(__shfl(i, 0));
		if (i >= n) break;
		const int t = uni(list[i]);
		const int l_query = uni(tasks[t].len), parent = uni(tasks[t].parent), n_iv = uni(task_n[t]);
		const uint32_t qoff = (uint32_t)uni((int)tasks[t].qoff);
		const long long po = pos ? uni64(pos_off[t]) : -1;
		RG_DBG_FILL(S, D);
		int status = rg_task<Store, true, DPT>(S, D, ix, sc, P, reads, l_query, parent, qoff, seeds_dense + uni64(task_off[t]), n_iv, po >= 0 ? pos + po : nullptr, lane, counters, gap_tab, ctg_tab, &X, t);
		if (status == 11) continue;
		status = rg_publish(S, t, status, out, out_cap, out_cursor, reg_off, reg_n, lane);
		if ((status == 8 || status == 2 || status == 3 || status == 4 || status == 6 || status == 10) && lane == 0) {
			next_list[atomicAdd(next_count, 1u)] = t;
			if (P.prof) atomicAdd(counters + 160 + status, 1ull);   // ($BSX_PHASES=2: why this tier hands a strand search on)
		}
	}
	RG_PF_FLUSH(D);
	RG_DBG_GUARDS("k_regions_mid");
}

// ---- K3 for the whole chunk ahead of the region kernels: the LF walks are pure pointer chasing, and run an order of
// magnitude faster with one walk per lane and thousands of waves in flight than inside the wave-per-task kernels.
// k_occ_expand lists the SA ranks of every occurrence each strand search will visit (lane per strand search),
// k_occ turns each rank into its reference position in place (lane per occurrence, bwt_sa, bwt.c:87-97).
#define OCC_MAX_PER_TASK 8192   // = RgHuge::SCAP: nothing on the device visits more
__global__ void __launch_bounds__(256)
k_occ_expand(const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n, int max_occ,
             unsigned long long *desc, unsigned long long desc_cap, unsigned long long *cursor, long long *pos_off, unsigned char *cls,
             int *early_list, unsigned int *early_count)
{
	const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (t >= n_tasks) return;
	const int n_iv = task_n[t];
	long long off = -1;
	unsigned char tier = 0;
	if (n_iv > 0) {
		const DevIntv *src = seeds_dense + task_off[t];
		unsigned long long tot = 0; int over = 0;
		for (int i = 0; i < n_iv; ++i) { const unsigned long long x2 = src[i].x2; tot += x2 > (unsigned long long)max_occ ? (unsigned long long)max_occ : x2; }   // the first max_occ of an over-represented interval
		// which tier's tables hold this strand search: decided here, where the sizes are known, so that the first tier does not start
		// what it would have to give up (chains and regions can still outgrow a tier: that is found out along the way)
		tier = (n_iv <= RgSmall::ICAP && tot <= (unsigned long long)RgSmall::SCAP) ? 0 : (n_iv <= RgMid::ICAP && tot <= (unsigned long long)RgMid::SCAP) ? 1 : 2;
		// what only the last HBM tier's tables hold (rg_task's own tests, statuses 8 and 2, of every tier before it) is listed for it here: that
		// launch -- a couple of thousand strand searches, as long as its longest -- then runs beside the other tiers instead of behind them
		if (early_list && (n_iv > RgBig::ICAP || tot > (unsigned long long)RgBig::SCAP)) { tier = 3; early_list[atomicAdd(early_count, 1u)] = t; }
		if (!over && tot > 0 && tot <= OCC_MAX_PER_TASK) {
			const unsigned long long base = atomicAdd(cursor, tot);
			if (base + tot <= desc_cap) {
				const unsigned long long par = (unsigned long long)(tasks[t].parent & 1) << 63;
				unsigned long long j = base;
				for (int i = 0; i < n_iv; ++i) {
					const unsigned long long x0 = src[i].x0, x2 = src[i].x2 > (unsigned long long)max_occ ? (unsigned long long)max_occ : src[i].x2;
					for (unsigned long long k = 0; k < x2; ++k) desc[j++] = (x0 + k) | par;
				}
				off = (long long)base;
			} else {
				// no room: the strand search is chained with inline LF walks instead (pos_off = -1).  k_occ still visits every slot up to
				// min(cursor, cap): what this task reserved of them is given rank 0, where a walk ends at once, instead of whatever an earlier
				// chunk left there
				for (unsigned long long j = base; j < desc_cap && j < base + tot; ++j) desc[j] = 0;
			}
		}
	}
	pos_off[t] = off;
	if (cls) cls[t] = tier;
}

__global__ void __launch_bounds__(256)
k_occ(DevIndex ix, unsigned long long *desc, unsigned long long desc_cap, const unsigned long long *cursor, unsigned long long *counters, const unsigned long long *start)
{
	unsigned long long n = *cursor;
	if (n > desc_cap) n = desc_cap;
	uint32_t lf = 0, calls = 0;
	// start: where this launch's ranks begin (what lies before was turned into positions by an earlier launch over the same pool)
	for (unsigned long long j = (start ? *start : 0ull) + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x) {
		const unsigned long long v = desc[j];
		desc[j] = (unsigned long long)rg_sa(ix, (int)(v >> 63), v & 0x7fffffffffffffffull, lf);
		++calls;
	}
	for (int off = 32; off > 0; off >>= 1) { lf += __shfl_down(lf, off); calls += __shfl_down(calls, off); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[2], (unsigned long long)lf); atomicAdd(&counters[3], (unsigned long long)calls); }
}

void launch_occ(hipStream_t st, int n_cu, const DevIndex &ix, const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off,
                const int *task_n, int max_occ, unsigned long long *desc, unsigned long long desc_cap, unsigned long long *cursor, long long *pos_off,
                unsigned long long *counters, unsigned char *cls, unsigned long long *start, int *early_list, unsigned int *early_count)
{
	if (start) (void)hipMemcpyAsync(start, cursor, 8, hipMemcpyDeviceToDevice, st);   // a later launch over the same pool: only the ranks it adds
	hipLaunchKernelGGL(k_occ_expand, dim3((n_tasks + 255) / 256), dim3(256), 0, st, tasks, n_tasks, seeds_dense, task_off, task_n, max_occ, desc, desc_cap, cursor, pos_off, cls, early_list, early_count);
	hipLaunchKernelGGL(k_occ, dim3(n_cu * 32), dim3(256), 0, st, ix, desc, desc_cap, cursor, counters, (const unsigned long long*)start);
}

// The index files sample the suffix array every 32nd rank (bwtindex.c:328,340), which makes bwt_sa a walk of 31 LF steps
// on average.  HBM has room for a much denser sample: at upload time every `intv`-th rank's position is computed once from
// the sparse samples (the value of SA[k] does not depend on the sampling), and all later lookups walk ~intv - 1 steps.
__global__ void __launch_bounds__(256)
k_sa_dense(DevIndex ix, int parent, unsigned int intv, unsigned long long n, unsigned long long *out)
{
	uint32_t lf = 0;
	for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x)
		out[j] = (unsigned long long)rg_sa(ix, parent, j * intv, lf);
}
void launch_sa_dense(hipStream_t st, int n_cu, const DevIndex &ix, int parent, unsigned int intv, unsigned long long n, unsigned long long *out)
{
	hipLaunchKernelGGL(k_sa_dense, dim3(n_cu * 32), dim3(256), 0, st, ix, parent, intv, n, out);
}

// The last HBM tier lasts as long as its longest strand search: a couple of thousand of them on a thousand waves, 40 ms each on average and 190 ms
// the longest -- when that one is taken off the cursor late, the launch is its 190 ms behind everything before it.  So the list is put in order of
// decreasing size first (the occurrences the strand search will visit, what k_occ_expand counted): longest first, the short ones fill in behind.
// One workgroup; ranks by counting (n is a few thousand).  Lists longer than the table are left as they are: they are bound by throughput anyway.
#define ORDER_CAP 6144
__global__ void __launch_bounds__(256)
k_order_list(int *list, const unsigned int *count, const DevIntv *seeds_dense, const long long *task_off, const int *task_n, int max_occ)
{
	__shared__ unsigned int cost[ORDER_CAP];
	__shared__ int task[ORDER_CAP];
	const int n = (int)*count;
	if (n < 2 || n > ORDER_CAP) return;
	for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
		const int t = list[i], n_iv = task_n[t];
		const DevIntv *src = seeds_dense + task_off[t];
		unsigned long long tot = 0;
		for (int k = 0; k < n_iv; ++k) { const unsigned long long x2 = src[k].x2; tot += x2 > (unsigned long long)max_occ ? (unsigned long long)max_occ : x2; }
		cost[i] = tot > 0xffffffffull ? 0xffffffffu : (unsigned int)tot; task[i] = t;
	}
	__syncthreads();
	for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
		const unsigned int c = cost[i];
		int r = 0;
		for (int j = 0; j < n; ++j) { const unsigned int cj = cost[j]; r += (cj > c || (cj == c && j < i)) ? 1 : 0; }
		list[r] = task[i];
	}
}
void launch_order_list(hipStream_t st, int *list, const unsigned int *count, const DevIntv *seeds_dense, const long long *task_off, const int *task_n, int max_occ)
{
	// (four waves: a workgroup of sixteen waited 13 ms on average for a CU with room for all of them in the pipelined trace; the ranks of 2 400 take 40 us either way)
	hipLaunchKernelGGL(k_order_list, dim3(1), dim3(256), 0, st, list, count, seeds_dense, task_off, task_n, max_occ);
}

size_t regions_slab_bytes(int tier) { return tier == 2 ? sizeof(RgBig) : sizeof(RgHuge); }
size_t c2r_hbm_slab_bytes(void) { return RgC2rH::slab_bytes() * C2R_WPB; }   // per workgroup of launch_c2r(.., long_reads = 3)

void launch_regions(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                    const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                    bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                    unsigned int *task_cursor, int *retry_list, unsigned int *retry_count, int quota, unsigned long long *counters,
                    const long long *pos_off, const unsigned long long *pos, const unsigned char *cls, const RgXPoolArg &XA, int long_reads)
{
	RgXPool X = rgx_pool(&XA);
	const int occ = (int)bsx_tune_long("regions_occ", 5);   // waves per SIMD the register allocation targets (the tables in LDS allow five workgroups per CU)
	if (long_reads)
		hipLaunchKernelGGL((k_regions<3, RgDpLiteL>), dim3(grid * (4 / RG_WPB)), dim3(64 * RG_WPB), 0, st, ix, sc, P, reads, tasks, n_tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, task_cursor, retry_list, retry_count, quota, counters, pos_off, pos, cls, X);
	else if (occ >= 5)
		hipLaunchKernelGGL((k_regions<5, RgDpLite>), dim3(grid * (4 / RG_WPB)), dim3(64 * RG_WPB), 0, st, ix, sc, P, reads, tasks, n_tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, task_cursor, retry_list, retry_count, quota, counters, pos_off, pos, cls, X);
	else if (occ >= 4)
		hipLaunchKernelGGL((k_regions<4, RgDpLite>), dim3(grid * (4 / RG_WPB)), dim3(64 * RG_WPB), 0, st, ix, sc, P, reads, tasks, n_tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, task_cursor, retry_list, retry_count, quota, counters, pos_off, pos, cls, X);
	else
		hipLaunchKernelGGL((k_regions<3, RgDpLite>), dim3(grid * (4 / RG_WPB)), dim3(64 * RG_WPB), 0, st, ix, sc, P, reads, tasks, n_tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, task_cursor, retry_list, retry_count, quota, counters, pos_off, pos, cls, X);
}

void launch_regions_mid(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                        const bsx_seed_task_t *tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                        bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                        const int *list, const unsigned int *count, unsigned int *cursor, int *next_list, unsigned int *next_count,
                        unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, const RgXPoolArg &XA, int quota, int long_reads)
{
	RgXPool X = rgx_pool(&XA);
	if (long_reads == 3)   // kilobase reads, the larger of the two table sizes
		hipLaunchKernelGGL((k_regions_mid<RgLongB, RgDpLiteL, 1, 1>), dim3(grid * 2), dim3(64), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, next_list, next_count, counters, pos_off, pos, X, quota);
	else if (long_reads == 4)   // reads of ordinary length: twice RgMid's tables, the tier behind it
		hipLaunchKernelGGL((k_regions_mid<RgMid2, RgDpLite, 1, 2>), dim3(grid * 2), dim3(64), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, next_list, next_count, counters, pos_off, pos, X, quota);
	else if (long_reads == 2)   // the larger tables for reads of ordinary length (the tier behind k_regions_mid<RgMid>)
		hipLaunchKernelGGL((k_regions_mid<RgLongB, RgDpLite, 1, 1>), dim3(grid * 2), dim3(64), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, next_list, next_count, counters, pos_off, pos, X, quota);
	else if (long_reads)   // tables for a kilobase read, one wave per workgroup
		hipLaunchKernelGGL((k_regions_mid<RgLongS, RgDpLiteL, 1, 1>), dim3(grid * 2), dim3(64), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, next_list, next_count, counters, pos_off, pos, X, quota);
	else
	hipLaunchKernelGGL((k_regions_mid<RgMid, RgDpLite, MID_WPB, 2>), dim3(grid * (2 / MID_WPB)), dim3(64 * MID_WPB), 0, st, /* `grid` counts pairs of waves */ ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
	                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, next_list, next_count, counters, pos_off, pos, X, quota);
}
void launch_c2r(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
                const RgXPoolArg &XA, bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                unsigned int *cursor, int *next_list, unsigned int *next_count, unsigned long long *counters, int quota, int long_reads, void *slab)
{
	RgXPool X = rgx_pool(&XA);
	if (long_reads == 4)   // chunks with long reads: what k_c2r<RgC2rL> declines
		hipLaunchKernelGGL((k_c2r<RgC2rHL, 1>), dim3(grid), dim3(64 * C2R_WPB), 0, st, ix, sc, P, reads, tasks, X, out, out_cap, out_cursor, reg_off, reg_n, cursor, next_list, next_count, counters, quota, (unsigned char*)slab);
	else if (long_reads == 3)   // ... and what that one declines too: the large tables in HBM (`slab`: c2r_hbm_slab_bytes() per wave of the grid)
		hipLaunchKernelGGL((k_c2r<RgC2rH, 2>), dim3(grid), dim3(64 * C2R_WPB), 0, st, ix, sc, P, reads, tasks, X, out, out_cap, out_cursor, reg_off, reg_n, cursor, next_list, next_count, counters, quota, (unsigned char*)slab);
	else if (long_reads == 2)   // ordinary reads with many regions or long seed lists: the strand searches k_c2r<RgC2r> declined (X names them)
		hipLaunchKernelGGL((k_c2r<RgC2rB, 2>), dim3(grid * (4 / C2R_WPB)), dim3(64 * C2R_WPB), 0, st, ix, sc, P, reads, tasks, X, out, out_cap, out_cursor, reg_off, reg_n, cursor, next_list, next_count, counters, quota, (unsigned char*)nullptr);
	else if (long_reads)
		hipLaunchKernelGGL((k_c2r<RgC2rL, 3>), dim3(grid * (4 / C2R_WPB)), dim3(64 * C2R_WPB), 0, st, ix, sc, P, reads, tasks, X, out, out_cap, out_cursor, reg_off, reg_n, cursor, next_list, next_count, counters, quota, (unsigned char*)nullptr);
	else
	hipLaunchKernelGGL((k_c2r<RgC2r, 4>), dim3(grid * (4 / C2R_WPB)), dim3(64 * C2R_WPB), 0, st, /* `grid` counts groups of four waves */ ix, sc, P, reads, tasks, X, out, out_cap, out_cursor, reg_off, reg_n, cursor, next_list, next_count, counters, quota, (unsigned char*)nullptr);
}
void launch_regions_slab(hipStream_t st, int tier, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                         const bsx_seed_task_t *tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                         bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                         const int *list, const unsigned int *count, unsigned int *cursor, void *slabs, int *next_list, unsigned int *next_count,
                         unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, const RgXPoolArg *XA)
{
	RgXPool X = rgx_pool(XA);
	// XA given: the tier stops after the chain filter and exports (chunks with long reads or an active seed-SW filter)
	if (tier == 2 && XA)
		hipLaunchKernelGGL((k_regions_slab<RgBig, true, RgDpLiteL>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, (RgBig*)slabs, next_list, next_count, counters, pos_off, pos, X);
	else if (tier == 2)
		hipLaunchKernelGGL((k_regions_slab<RgBig, false, RgDp>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, (RgBig*)slabs, next_list, next_count, counters, pos_off, pos, X);
	else if (XA)
		hipLaunchKernelGGL((k_regions_slab<RgHuge, true, RgDpLiteL>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, (RgHuge*)slabs, next_list, next_count, counters, pos_off, pos, X);
	else
		hipLaunchKernelGGL((k_regions_slab<RgHuge, false, RgDp>), dim3(grid), dim3(256), 0, st, ix, sc, P, reads, tasks, seeds_dense, task_off, task_n,
		                   out, out_cap, out_cursor, reg_off, reg_n, list, count, cursor, (RgHuge*)slabs, next_list, next_count, counters, pos_off, pos, X);
}
size_t seedsw_job_bytes(void) { return sizeof(SswJob) + sizeof(int); }   // a job and its score
void launch_seedsw(hipStream_t st, int grid, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
                   const RgXPoolArg &XA, unsigned int *cursor, unsigned int *count_cursor, void *jobs, unsigned int job_cap, unsigned long long *counters)
{   // cursor: k_seedsw_prep's; count_cursor: [0] the job count [1] k_seedsw_apply's cursor (zeroed by the caller)
	RgXPool X = rgx_pool(&XA);
	SswJob *J = (SswJob*)jobs;
	int *scores = (int*)(J + job_cap);
	{ // k_swl16 keeps two jobs' scores in the 16-bit halves of a register (wrapping adds, signed 16-bit maxima) and its profile as biased bytes:
	  // exact while the largest score of a job -- at most 199 columns of the matrix's maximum -- plus the bias and the gap terms added to it
	  // stays below 2^15, and a biased matrix entry fits a byte.  Scoring options beyond that (an -A in the hundreds) leave the job list empty:
	  // every seed then takes the no-room path of k_seedsw_apply (sw_pass: a wavefront per alignment in 32-bit arithmetic), which is exact.
		int mx = 0, bias = 0;
		for (int i = 0; i < 25; ++i) {
			mx = sc.ctmat[i] > mx ? sc.ctmat[i] : mx; mx = sc.gamat[i] > mx ? sc.gamat[i] : mx;
			bias = -(int)sc.ctmat[i] > bias ? -(int)sc.ctmat[i] : bias; bias = -(int)sc.gamat[i] > bias ? -(int)sc.gamat[i] : bias;
		}
		const int oe = (sc.o_del + sc.e_del > sc.o_ins + sc.e_ins ? sc.o_del + sc.e_del : sc.o_ins + sc.e_ins), e = sc.e_del > sc.e_ins ? sc.e_del : sc.e_ins;
		if (199LL * mx + 7LL * 25 * e + 2LL * oe + bias >= 32768 || mx + bias > 255 || oe >= 32768) job_cap = 0;
	}
	hipLaunchKernelGGL(k_seedsw_prep, dim3(grid), dim3(64), 0, st, ix, tasks, X, cursor, J, job_cap, count_cursor);
	if (sc.o_del == sc.o_ins && sc.e_del == sc.e_ins) hipLaunchKernelGGL(k_swl16<true>, dim3(n_cu * 8), dim3(256), 0, st, ix, sc, reads, (const SswJob*)J, (const unsigned int*)count_cursor, job_cap, scores);
	else hipLaunchKernelGGL(k_swl16<false>, dim3(n_cu * 8), dim3(256), 0, st, ix, sc, reads, (const SswJob*)J, (const unsigned int*)count_cursor, job_cap, scores);
	hipLaunchKernelGGL(k_seedsw_apply, dim3(grid), dim3(64), 0, st, ix, sc, P, reads, tasks, X, count_cursor + 1, (const int*)scores, counters);
}
int regions_long_max_query(void) { return RG_QCAP_LONG; }
