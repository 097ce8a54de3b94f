// seed_tab.hpp -- K1+K2 over a table of k-mer intervals: the per-lane state machine of k_seed.
//
// What the machine computes is mem_collect_intv (lib/aln/memchain.c:50-106): three passes over bwt_smem1a / bwt_seed_strategy1
// (lib/aln/bwt.c:307-396), interval for interval.  How it gets there is this device's own:
//
// (1) A converted read has three letters (C>T leaves A,G,T; G>A leaves A,C,T), and the bi-interval of a string is a property of the
//     string, not of the order in which it was extended (bwt_extend, bwt.c:278-293, keeps x0/x1/x2 exact in both directions; checked
//     against the FM index on millions of substrings, tests/test_kernel_logic_host.py).  So the bi-intervals of ALL strings of up to K
//     letters sit in a table in HBM (level L: 3^L entries of 16 bytes; K = 18 for an hg38-sized text: 9.3 GB per converted index of the
//     288 GB), built once by K rounds of bwt_extend, and any substring of a read of up to K letters costs one 16-byte load instead of
//     up to K dependent pairs of FM blocks.
//
// (2) bwt_smem1a's backward sweep (bwt.c:344-365) extends every entry of the forward list by one base per row: for a read that
//     does not belong to the strand it is searched on that is ~20 entries x ~20 rows = ~210 bwt_extend per call, seven calls per
//     read.  Its result is a function of the matrix s(a, b) = interval size of read[a, b): with b running over the ends on the forward
//     list (longest first) and a(b) = the leftmost start for which s(a, b) >= min_intv, the SMEMs are exactly the (a(b), b) with
//     a(b) smaller than every a(b') of a longer b' -- entries that stop at the same row as a longer one are the ones the sweep
//     drops as contained (curr->n > 0), entries whose size equals a longer one's have the same a(b) (equal sizes stay equal under
//     further extension) and are the ones the sweep drops as duplicates (bwt.c:357-360).  The machine therefore walks ENTRY BY ENTRY:
//     for each b it needs s only around a(b), and a(b) only moves left.  Entries within the table's reach start from one lookup at
//     read[max(a_prev - 1, b - K), b) -- dead there means not an SMEM -- and walk left from it (lookups while the window has at
//     most K letters, bwt_extend beyond); entries longer than K start from the interval the forward extension stored.  An entry
//     whose size reaches the size the previous one ended with is dropped (the duplicate rule), and the call ends with the first
//     entry that reaches the start of the read or an N (bwt.c:346,362: nothing survives that row).
//
// (3) The forward extension (bwt.c:324-339) starts from the table's entry for read[x0, x0 + K) when that is still alive, and
//     bwt_seed_strategy1 (bwt.c:376-396) from the entry for its first min(K, min_len) letters: no test is made before that length.
//
// One request per lane per trip of the wave loop, as before: an FM extension (one or two 64-byte blocks, fetched by the wave as a
// whole) or a table entry (16 bytes, loaded by the lane itself), both in flight together.
#pragma once
#include "seed_core.hpp"

struct SeedTab {              // the interval tables of the two converted indices: t[parent][((3^L - 3) >> 1) + key], L = 1..K
	const SeedEnt *t[2];
	int32_t K;                // 0: no table (every step is an FM extension)
};
// key of a string: its letters as base-3 digits, the first letter the most significant one; digit of a converted base
BSX_HD uint32_t seed_digit(int b, int parent) { return (uint32_t)(b - (b > (parent ? 0 : 1))); }   // parent {A,G,T} -> 0,1,2; daughter {A,C,T} -> 0,1,2
BSX_HD int seed_letter(uint32_t d, int parent) { return parent ? (d ? (int)d + 1 : 0) : (d == 2 ? 3 : (int)d); }
BSX_HD uint64_t seed_tab_entries(int K) { uint64_t n = 0, p = 1; for (int l = 1; l <= K; ++l) { p *= 3; n += p; } return n; }
// the deepest table worth its memory for a text of n symbols: 3^K <= n / 8 (the entries of the last level then stand for ~8 suffixes
// each, and the table is at most ~1.5 x 16 x n / 8 = 3 bytes per symbol next to the index's 2.5), at most 18
BSX_HD int seed_tab_depth(uint64_t n_symbols) { int k = 0; uint64_t p = 3; while (k < 18 && p <= n_symbols / 8) { ++k; p *= 3; } return k < 2 ? 0 : k; }

struct SeedLane2 {
	// task
	const uint8_t *q;
	const uint32_t *qlds;
	int32_t len, parent;
	// scratch (lane-interleaved slab, as SeedLane)
	SeedEnt *bufA;            // forward-list entries longer than the table reaches (end of match in `hi`), in the order they were pushed
	DevIntv *mem;             // the SMEMs found so far: this lane's entry 0, entry k at k * mstride bytes (the wave's slab, lane-interleaved, or the
	uint32_t mstride;         // strand search's own stretch of the output)
	int32_t list_cap, mem_cap, stride, lane;
	// machine
	int32_t state, ret_state;
	int32_t pass_x, k2, old_n;
	int32_t x0, min_intv, ret;
	int32_t i;                // forward position (SMEM forward extension, LAST-like walk)
	int32_t nlist;            // stored forward entries still to be walked (taken from the top: longest first)
	int32_t kcall;            // this call's forward extension started from the table at this length (0: from its first base)
	int32_t b, a, a_prev;     // the entry being walked is read[a, b); a_prev: the leftmost start reached by the longer entries (x0 + 1: none yet)
	int32_t edge;             // the walk stopped at the start of the read or at an N
	uint64_t sig_prev;        // size of the entry that reached a_prev, at a_prev
	uint32_t key, pw;         // key of read[ka, kb) and 3^(kb - ka); pw == 0: no key.  The window of a walk's lookups, and what the next
	int32_t ka, kb;           // entry's first window is made from by dropping and adding a letter or two (seed2_slide)
	int32_t mem_n, overflow;
	uint64_t resplit0, resplit1;   // which of pass 1's first 128 SMEMs pass 2 re-seeds from (two scalars: an indexed array would live in scratch memory)
	DevIntv ik;               // forward: interval of read[x0, i); backward: of read[a, b)
	// request
	int32_t ext_back, ext_c, ext_which;   // FM: bwt_extend of ik
	uint32_t tab_idx;                     // lookup: entry of t[parent]
	uint32_t n_slow, n_fast, n_look;
};

enum { SQ_NONE = 0, SQ_FM = 1, SQ_TAB = 2 };
enum { ST_DONE = 0, ST_P1, ST_P2, ST_P3, ST_SMEM_BEGIN, ST_SMEM_END, ST_FPROBE_POST, ST_FWD, ST_FWD_POST, ST_NEXT, ST_ENT_POST, ST_WALK, ST_WALK_POST,
       ST_FINISH, ST_S1, ST_S1_POST, ST_S1J_POST };

// The read in LDS: bisulfite-converted (bseq_bsconvert, lib/aln/bwamem.c:161-178) and stored as the table's base-3 digits, two bits
// a base (3 = N), sixteen bases a word, lane-interleaved: qlds[(i >> 4) * 64].  Reads that do not fit are read where they lie.
BSX_HD uint32_t seed2_digit_of(int nt4, int parent) { return ((parent ? 0x398u : 0x384u) >> ((nt4 > 4 ? 4 : nt4) << 1)) & 3u; }   // parent A,C,G,T,N -> 0,2,1,2,3; daughter -> 0,1,0,2,3
BSX_HD uint32_t seed2_qdigit(const SeedLane2 &L, int i)
{
	if (L.qlds) return (L.qlds[(i >> 4) << 6] >> ((i & 15) << 1)) & 3u;
	return seed2_digit_of(L.q[i], L.parent);
}
BSX_HD int seed2_qbase(const SeedLane2 &L, int i)   // the converted base: 0..3, 4 = N
{
	const uint32_t d = seed2_qdigit(L, i);
	return d == 3 ? 4 : seed_letter(d, L.parent);
}
BSX_HD DevIntv &seed2_mem_at(const SeedLane2 &L, int idx)
{
	return *reinterpret_cast<DevIntv*>(reinterpret_cast<char*>(L.mem) + (size_t)((uint32_t)idx * L.mstride));
}
BSX_HD SeedEnt &seed2_list_at(const SeedLane2 &L, int idx)
{
	return *reinterpret_cast<SeedEnt*>(reinterpret_cast<char*>(L.bufA) + (uint32_t)((uint32_t)(idx * L.stride + L.lane) * (uint32_t)sizeof(SeedEnt)));
}
BSX_HD void seed2_emit(SeedLane2 &L, const DevIntv &m, int beg, int end)
{
	if (L.mem_n < L.mem_cap) { DevIntv o = m; o.info = (uint64_t)(uint32_t)beg << 32 | (uint32_t)end; seed2_mem_at(L, L.mem_n) = o; }
	else L.overflow = 1;
	++L.mem_n;
}
BSX_HD void seed2_lane_begin(SeedLane2 &L)
{
	L.pass_x = 0; L.mem_n = 0; L.overflow = 0; L.n_slow = L.n_fast = L.n_look = 0;
	L.resplit0 = L.resplit1 = 0;
	L.state = ST_P1;
}
// key of read[from, from + n) for the largest n <= max_n that stays inside the read and before the next N; n is returned, key/pw = 3^n set
BSX_HD int seed2_scan(const SeedLane2 &L, int from, int max_n, uint32_t &key, uint32_t &pw)
{
	uint32_t k = 0, p = 1;
	int n = 0;
	if (max_n > L.len - from) max_n = L.len - from;
	if (L.qlds) { // a word of sixteen digits at a time
		while (n < max_n) {
			const int pos = from + n;
			uint32_t w = L.qlds[(pos >> 4) << 6] >> ((pos & 15) << 1);
			int m = 16 - (pos & 15);
			if (m > max_n - n) m = max_n - n;
			bool stop = false;
			for (int j = 0; j < m; ++j, w >>= 2) {
				const uint32_t d = w & 3u;
				if (d == 3) { stop = true; break; }
				k = k * 3u + d; p *= 3u; ++n;
			}
			if (stop) break;
		}
	} else {
		for (; n < max_n; ++n) {
			const uint32_t d = seed2_digit_of(L.q[from + n], L.parent);
			if (d == 3) break;
			k = k * 3u + d; p *= 3u;
		}
	}
	key = k; pw = p;
	return n;
}
// the key moved from read[ka, kb) to read[at, nb) (nb <= kb), a letter at a time: consecutive entries differ by one letter at either end
BSX_HD void seed2_slide(SeedLane2 &L, int at, int nb)
{
	const uint32_t INV3 = 0xAAAAAAABu;   // 3 * INV3 == 1 (mod 2^32): exact division of a multiple of three
	if (L.pw == 0 || L.kb < nb || L.ka >= nb || L.kb - nb > 4 || L.ka - at > 4 || at - L.ka > 4) { seed2_scan(L, at, nb - at, L.key, L.pw); L.ka = at; L.kb = nb; return; }
	while (L.kb > nb) { --L.kb; L.key = (L.key - seed2_qdigit(L, L.kb)) * INV3; L.pw *= INV3; }
	while (L.ka > at) { --L.ka; L.key += seed2_qdigit(L, L.ka) * L.pw; L.pw *= 3u; }
	while (L.ka < at) { L.pw *= INV3; L.key -= seed2_qdigit(L, L.ka) * L.pw; ++L.ka; }
}
BSX_HD uint32_t seed2_tab_at(uint32_t key, uint32_t pw) { return ((pw - 3u) >> 1) + key; }   // entry of a string with key `key` and 3^length = pw
BSX_HD void seed2_req_fm(SeedLane2 &L, int back, int base)
{
	L.ext_back = back; L.ext_c = back ? base : 3 - base; L.ext_which = back ? 0 : 1;
}

// Runs the machine until it needs memory: SQ_FM (bwt_extend of L.ik: ext_back/ext_c/ext_which) or SQ_TAB (entry L.tab_idx of the table of
// L.parent), the result goes to seed2_post; SQ_NONE: the strand search is done.
// The states are visited in the order a strand search moves through them, so that a lane normally reaches its next request in one
// pass (an entry ends -> the next entry's lookup; a call ends -> pass control -> the next call's probe); a second pass is the exception.
BSX_HD int seed2_advance(SeedLane2 &L, const DevIndex &ix, const SeedParams &P, int K)
{
	int req = SQ_NONE;
	for (;;) {
		if (L.state == ST_FINISH) { // read[a, b) is as far left as this entry goes (bwt.c:350-355)
			if (L.a < L.a_prev) {
				if (L.b - L.a >= P.min_seed_len) {
					if (L.ret_state == ST_P1 && L.b - L.a >= P.split_len && L.ik.x2 <= (uint64_t)P.split_width && L.mem_n < 128) // one pass 2 comes back to (memchain.c:79)
						{ const uint64_t bit = 1ull << (L.mem_n & 63); if (L.mem_n < 64) L.resplit0 |= bit; else L.resplit1 |= bit; }
					seed2_emit(L, L.ik, L.a, L.b);
				}
				L.a_prev = L.a;
			}
			L.sig_prev = L.ik.x2;
			L.state = L.edge ? ST_SMEM_END : ST_NEXT;
		}
		if (L.state == ST_NEXT) { // the next shorter entry
			while (L.nlist > 0) { // one the forward extension stored: it starts at x0 (inside the stretch the longer entries covered)
				const SeedEnt e = seed2_list_at(L, --L.nlist);
				L.ik = seed_unpack(e); L.b = (int)(e.hi >> 8); L.ik.info = 0;
				L.a = L.x0;
				if (L.ik.x2 == L.sig_prev) continue;   // as large as the longer entry at its leftmost start: a duplicate from here on
				L.state = ST_WALK;
				break;
			}
			if (L.state == ST_NEXT) {
				int nb = L.b - 1;
				if (nb > L.x0 + L.kcall) nb = L.x0 + L.kcall;
				if (nb <= L.x0) L.state = ST_SMEM_END;
				else {
					L.b = nb;
					int at = L.a_prev - 1;
					if (at < nb - K) at = nb - K;
					L.a = at;
					seed2_slide(L, at, nb);   // read[a_prev - 1, x0] holds no N: the walks that got there extended by every base of it
					L.tab_idx = seed2_tab_at(L.key, L.pw);
					L.state = ST_ENT_POST;
					req = SQ_TAB;
				}
			}
		}
		if (L.state == ST_SMEM_END) {
			if (L.ret_state == ST_P1) L.pass_x = L.ret;
			L.state = L.ret_state;
		}
		if (L.state == ST_P1) { // pass 1: SMEMs from every position (memchain.c:65-73)
			while (L.pass_x < L.len && seed2_qdigit(L, L.pass_x) == 3) ++L.pass_x;
			if (L.pass_x >= L.len) { L.old_n = L.mem_n; L.k2 = 0; L.state = ST_P2; }
			else { L.x0 = L.pass_x; L.min_intv = P.start_width; L.ret_state = ST_P1; L.state = ST_SMEM_BEGIN; }
		}
		if (L.state == ST_P2) { // pass 2: re-seed from the middle of long, rare SMEMs (memchain.c:76-85); the first 128 of pass 1 were marked when they were emitted
			for (;;) {
				if (L.k2 < 128) { // the next marked one
					uint64_t m = L.k2 < 64 ? L.resplit0 >> L.k2 : 0;
					if (m) { while (!(m & 1)) { m >>= 1; ++L.k2; } }
					else {
						m = L.resplit1 >> (L.k2 < 64 ? 0 : L.k2 - 64);
						if (L.k2 < 64) L.k2 = 64;
						if (m) { while (!(m & 1)) { m >>= 1; ++L.k2; } } else L.k2 = 128;
					}
				}
				if (L.k2 >= L.old_n) { L.pass_x = 0; L.state = P.max_mem_intv > 0 ? ST_P3 : ST_DONE; break; }
				const int kk = L.k2++;
				if (kk < L.mem_cap) {
					const DevIntv p = seed2_mem_at(L, kk);
					const int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
					if (end - start < P.split_len || p.x2 > (uint64_t)P.split_width) continue;
					L.x0 = (start + end) >> 1; L.min_intv = (int)(p.x2 + 1); L.ret_state = ST_P2; L.state = ST_SMEM_BEGIN;
					break;
				}
			}
		}
		if (L.state == ST_P3) { // pass 3: LAST-like forward-only seeds (memchain.c:88-103), bwt_seed_strategy1 (bwt.c:376-396)
			while (L.pass_x < L.len && seed2_qdigit(L, L.pass_x) == 3) ++L.pass_x;
			if (L.pass_x >= L.len) L.state = ST_DONE;
			else {
				L.x0 = L.pass_x;
				int n = 0;
				if (K >= 2) { // no test is made before the match has min_len letters (bwt.c:387): start from the table's entry for the first min(K, min_len)
					n = seed2_scan(L, L.x0, K < P.min_seed_len ? K : P.min_seed_len, L.key, L.pw);
					L.ka = L.x0; L.kb = L.x0 + n;
				}
				if (n >= 2) { L.tab_idx = seed2_tab_at(L.key, L.pw); L.i = L.x0 + n; L.state = ST_S1J_POST; req = SQ_TAB; }
				else { seed_set_intv(ix, L.parent, seed2_qbase(L, L.x0), L.ik); L.i = L.x0 + 1; L.state = ST_S1; }
			}
		}
		if (L.state == ST_SMEM_BEGIN) { // bwt_smem1a prologue (bwt.c:313-322); max_intv is always 0 here
			if (seed2_qdigit(L, L.x0) == 3) { L.ret = L.x0 + 1; L.state = ST_SMEM_END; }
			else {
				if (L.min_intv < 1) L.min_intv = 1;
				L.nlist = 0; L.kcall = 0;
				int n = 0;
				if (K >= 2) { n = seed2_scan(L, L.x0, K, L.key, L.pw); L.ka = L.x0; L.kb = L.x0 + n; }
				if (n >= 2) { L.tab_idx = seed2_tab_at(L.key, L.pw); L.i = L.x0 + n; L.state = ST_FPROBE_POST; req = SQ_TAB; }
				else { seed_set_intv(ix, L.parent, seed2_qbase(L, L.x0), L.ik); L.i = L.x0 + 1; L.pw = 0; L.state = ST_FWD; }
			}
		}
		if (L.state == ST_FWD) { // forward extension through the complementary index (bwt.c:324-339); L.ik = interval of read[x0, i)
			const int b = L.i < L.len ? seed2_qbase(L, L.i) : 4;
			if (b < 4) { seed2_req_fm(L, 0, b); L.state = ST_FWD_POST; req = SQ_FM; }
			else { // end of the read or an N: read[x0, i) is the longest entry and the first to be walked; it stays in L.ik
				L.ret = L.i; L.b = L.i; L.a = L.x0; L.a_prev = L.x0 + 1; L.sig_prev = 0;   // (if it is read[x0, x0 + kcall), the probe's key is this entry's)
				L.state = ST_WALK;
			}
		}
		if (L.state == ST_WALK) { // extend read[a, b) to the left by one base
			const int c = L.a > 0 ? seed2_qbase(L, L.a - 1) : 4;
			if (c > 3) { L.edge = 1; L.state = ST_FINISH; }
			else {
				L.ext_c = c;
				if (L.pw != 0 && L.ka == L.a && L.kb == L.b && L.b - L.a < K) { // the key is this window's, and the longer one is still within the table
					L.tab_idx = seed2_tab_at(L.key + seed_digit(c, L.parent) * L.pw, L.pw * 3u);
					L.ext_back = 2;
					req = SQ_TAB;
				} else { seed2_req_fm(L, 1, c); req = SQ_FM; }
				L.state = ST_WALK_POST;
			}
		}
		if (L.state == ST_S1) { // bwt_seed_strategy1 loop (bwt.c:384-394)
			const int b = L.i < L.len ? seed2_qbase(L, L.i) : 5;
			if (b < 4) { seed2_req_fm(L, 0, b); L.state = ST_S1_POST; req = SQ_FM; }
			else { L.pass_x = b == 5 ? L.len : L.i + 1; L.state = ST_P3; }
		}
		if (req != SQ_NONE || L.state == ST_DONE) return req;
	}
}

// Consume the result of the request: `ok` = the extended interval (FM) or the table's entry (x0, x1, x2; info 0).
BSX_HD void seed2_post(SeedLane2 &L, const DevIntv &ok, const DevIndex &ix, const SeedParams &P, int K)
{
	switch (L.state) {
	case ST_FPROBE_POST: // entry of read[x0, i)
		if (ok.x2 >= (uint64_t)L.min_intv) { L.ik = ok; L.kcall = L.i - L.x0; }
		else { // fewer than min_intv already: where the extension ends is found the slow way (the first bases may differ in size from the test's view)
			seed_set_intv(ix, L.parent, seed2_qbase(L, L.x0), L.ik);
			L.i = L.x0 + 1; L.pw = 0;
		}
		L.state = ST_FWD;
		break;
	case ST_FWD_POST:
		if (ok.x2 != L.ik.x2) { // interval size changed (bwt.c:329-333)
			if (ok.x2 < (uint64_t)L.min_intv) { // read[x0, i) is the longest entry
				L.ret = L.i; L.b = L.i; L.a = L.x0; L.a_prev = L.x0 + 1; L.sig_prev = 0; L.edge = 0;
				L.state = ST_WALK;
				break;
			}
			if (L.i - L.x0 > L.kcall) { // an entry the table cannot give back: keep it (end of the match in `hi`)
				if (L.nlist < L.list_cap) { DevIntv o = L.ik; o.info = (uint64_t)L.i; seed2_list_at(L, L.nlist) = seed_pack(o); ++L.nlist; }
				else L.overflow = 1;
			}
		}
		L.ik = ok; L.ik.info = 0;
		++L.i; L.state = ST_FWD;
		break;
	case ST_ENT_POST: // entry of read[a, b), a = max(a_prev - 1, b - K)
		L.ik = ok; L.edge = 0;
		if (L.a < L.a_prev) { // one base beyond the longer entries: alive there or not an SMEM
			L.state = ok.x2 >= (uint64_t)L.min_intv ? ST_WALK : ST_NEXT;
		} else L.state = ok.x2 == L.sig_prev ? ST_NEXT : ST_WALK;   // inside their stretch: alive for sure, dropped once it is no larger than they were
		break;
	case ST_WALK_POST:
		if (ok.x2 < (uint64_t)L.min_intv) { L.edge = 0; L.state = ST_FINISH; break; }
		if (L.ext_back == 2) { L.key += seed_digit(L.ext_c, L.parent) * L.pw; L.pw *= 3u; --L.ka; }
		--L.a; L.ik = ok; L.ik.info = 0;
		L.state = (L.a >= L.a_prev && ok.x2 == L.sig_prev) ? ST_NEXT : ST_WALK;
		break;
	case ST_S1J_POST: // entry of read[x0, i): as if the walk had come this far (a dead entry is all zeroes and stays dead)
		L.ik = ok; L.state = ST_S1;
		break;
	case ST_S1_POST:
		if (ok.x2 < (uint64_t)P.max_mem_intv && L.i - L.x0 >= P.min_seed_len) { // bwt.c:387-391
			if (ok.x2 > 0) seed2_emit(L, ok, L.x0, L.i + 1);
			L.pass_x = L.i + 1; L.state = ST_P3;
		} else { L.ik = ok; L.ik.info = 0; ++L.i; L.state = ST_S1; }
		break;
	default: break;
	}
}

// ---- the table.  Level 1 = bwt_set_intv of the three letters; an entry of level L + 1 = bwt_extend (forward, through the complementary
// index: what bwt_smem1a's forward extension does) of the entry of its first L letters.  Dead entries (x2 == 0) are all zeroes: an FM
// extension of one asks for no block (k = l = -1) and stays dead.
// children of one entry from the four-symbol counts of its extension (tk/tl = bwt_2occ4 at xa - 1 and xa - 1 + x2 in index !parent)
BSX_HD void seed_tab_children(const DevIntv &p, const uint64_t tk[4], const uint64_t tl[4], uint64_t primary, const uint64_t L2[5], int parent, SeedEnt out[3])
{
	const uint64_t xa = p.x1, xb = p.x0, x2 = p.x2;
	const uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1], s0 = tl[0] - tk[0];
	const uint64_t b3 = xb + ((xa <= primary && xa + x2 - 1 >= primary) ? 1 : 0);
	const uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const uint64_t bb[4] = {b0, b1, b2, b3}, ss[4] = {s0, s1, s2, s3};
	for (uint32_t d = 0; d < 3; ++d) {
		const int c = 3 - seed_letter(d, parent);   // forward extension by a letter = backward extension of the other strand by its complement
		DevIntv o; o.x0 = bb[c]; o.x1 = L2[c] + 1 + tk[c]; o.x2 = ss[c]; o.info = 0;
		if (o.x2 == 0) { o.x0 = o.x1 = 0; }
		out[d] = seed_pack(o);
	}
}
