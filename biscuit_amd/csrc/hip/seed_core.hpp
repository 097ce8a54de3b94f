// seed_core.hpp -- per-lane state machine for K1+K2: the three seeding passes of mem_collect_intv
// (lib/aln/memchain.c:50-106) over bwt_smem1a / bwt_seed_strategy1 (lib/aln/bwt.c:307-396).
//
// MI355X shape: one lane owns one (read, strand) search.  The reference's nested loops are turned
// inside out so that every trip of the wave-level loop performs exactly ONE bwt_extend per active
// lane -- the only step that touches HBM (one or two random 64-byte FM blocks).  All 64 lanes of a
// wave, whatever phase each is in (forward walk, backward sweep, LAST-like pass), issue their
// gathers together; thousands of waves in flight hide the dependent-gather latency.  Per-lane
// interval lists live in a private slab of global scratch, not in registers.
#pragma once
#include "dev_common.hpp"

struct SeedParams {           // copied from bsx_opt_t
	int32_t min_seed_len;
	int32_t split_len;        // (int)(min_seed_len * split_factor + .499), computed on the host
	int32_t split_width;
	int32_t max_mem_intv;
	int32_t start_width;      // 2 with BSX_F_SELF_OVLP else 1
};

// An entry of the interval lists (bwtintv_t, lib/aln/bwt.h:80-82) in 16 bytes instead of 32: x0, x1, x2 are ranks and sizes below 2^34
// (texts of up to 8.5 G symbols), `info` inside bwt_smem1a is the end of the match on the read (bwt.c:320,337).  The lists are written once
// per surviving interval per row of the backward sweep and were a quarter of the kernel's HBM traffic at 32 bytes an entry.
struct SeedEnt { uint32_t x0, x1, x2, hi; };   // hi: bits 32-33 of x0 | of x1 << 2 | of x2 << 4 | info << 8
BSX_HD SeedEnt seed_pack(const DevIntv &v)
{
	SeedEnt e;
	e.x0 = (uint32_t)v.x0; e.x1 = (uint32_t)v.x1; e.x2 = (uint32_t)v.x2;
	e.hi = (uint32_t)(v.x0 >> 32) | (uint32_t)(v.x1 >> 32) << 2 | (uint32_t)(v.x2 >> 32) << 4 | (uint32_t)v.info << 8;
	return e;
}
BSX_HD DevIntv seed_unpack(const SeedEnt &e)
{
	DevIntv v;
	v.x0 = (uint64_t)(e.hi & 3u) << 32 | e.x0; v.x1 = (uint64_t)(e.hi >> 2 & 3u) << 32 | e.x1; v.x2 = (uint64_t)(e.hi >> 4 & 3u) << 32 | e.x2;
	v.info = e.hi >> 8;
	return v;
}

// One lane's working state.  Lists: A and B are the prev/curr interval lists of bwt_smem1a.
// The forward list is written downwards from the top of its buffer, which yields the reversed
// order ("longest match first") the backward sweep wants without a reversal pass.
struct SeedLane {
	// task
	const uint8_t *q;         // raw read (nt4); converted on the fly
	const uint32_t *qlds;     // optional: the converted read packed 8 bases per word at qlds[(i>>3)*64] (LDS, lane-interleaved)
	int32_t len, parent;
	// scratch
	SeedEnt *bufA;                // the interval list of the walk in progress: element i of this lane lives at base[i * stride + lane], base
	                              // being the same for the whole wave (kept in scalar registers; the lane's part is a 32-bit offset).  ONE buffer serves the
	                              // forward list and every row of the backward sweep: a row's survivors are written over the row they come
	                              // from (survivor m <= j is stored after entry j has been read, the forward list sits at the top)
	DevIntv *mem;                 // the SMEMs found so far, same layout
	int32_t list_cap, mem_cap;
	int32_t stride, lane;     // 64 and the lane number on the GPU: the 64 lanes' i-th entries are contiguous; 1 and 0 on the host
	// machine
	int32_t state, ret_state;
	int32_t pass_x;           // pass-1 / pass-3 scan position
	int32_t k2, old_n;        // pass-2 cursor
	int32_t x0, min_intv;     // current smem1 call
	int32_t i, j, c;
	int32_t nprev, ncurr;     // list sizes
	int32_t prev_off;         // start offset of prev inside its buffer
	int32_t last_beg;         // start of the last SMEM emitted by this call, -1 if none
	int32_t ret;              // return value of the current smem1 call
	int32_t mem_n;
	int32_t overflow;
	uint64_t last_x2;         // interval size of the last survivor pushed in this row
	DevIntv ik;
	// pending extend request
	int32_t ext_back, ext_c, ext_which;   // ext_which: 0 = own index, 1 = complementary index
	DevIntv ext_in;
	SeedEnt next_in;          // prev[j+1], requested one step ahead so that its latency overlaps the FM gathers
	SeedEnt head;             // entry 0 of the list being built (the forward list's latest push, a backward row's first survivor).
	                          // It is never stored: the next row reads it from here.  Most backward rows have a single survivor,
	                          // so most extensions write nothing at all to the lists in scratch
	int32_t have_next;
	uint32_t n_slow, n_fast;
};

enum { SD_DONE = 0, SD_P1, SD_SMEM_BEGIN, SD_FWD, SD_FWD_POST, SD_FWD_DONE, SD_BWD_ELEM, SD_BWD_POST,
       SD_SMEM_END, SD_P2, SD_P3, SD_S1, SD_S1_POST };

BSX_HD int seed_qbase(const SeedLane &L, int i)
{
	if (L.qlds) return (int)((L.qlds[(i >> 3) << 6] >> ((i & 7) << 2)) & 15u);
	int b = L.q[i];
	return L.parent ? (b == 1 ? 3 : b) : (b == 2 ? 0 : b);   // bseq_bsconvert, lib/aln/bwamem.c:161-178
}

// element idx of this lane in its slab: a wave-uniform base plus a byte offset that fits 32 bits
BSX_HD DevIntv &seed_mem_at(const SeedLane &L, int idx)
{
	return *reinterpret_cast<DevIntv*>(reinterpret_cast<char*>(L.mem) + (uint32_t)((uint32_t)(idx * L.stride + L.lane) * (uint32_t)sizeof(DevIntv)));
}
BSX_HD SeedEnt &seed_list_at(const SeedLane &L, int idx)
{
	return *reinterpret_cast<SeedEnt*>(reinterpret_cast<char*>(L.bufA) + (uint32_t)((uint32_t)(idx * L.stride + L.lane) * (uint32_t)sizeof(SeedEnt)));
}

BSX_HD void seed_emit(SeedLane &L, const DevIntv &m, int beg, int end)
{
	if (end - beg < 0) return;
	if (L.mem_n < L.mem_cap) { DevIntv o = m; o.info = (uint64_t)(uint32_t)beg << 32 | (uint32_t)end; seed_mem_at(L, L.mem_n) = o; }
	else L.overflow = 1;
	++L.mem_n;
}

BSX_HD void seed_lane_begin(SeedLane &L)
{
	L.pass_x = 0; L.mem_n = 0; L.overflow = 0; L.n_slow = L.n_fast = 0;
	L.state = SD_P1;
}

BSX_HD void seed_set_intv(const DevIndex &ix, int parent, int c, DevIntv &ik)   // bwt_set_intv, lib/aln/bwt.h:105
{
	const uint64_t l = dev_ix_L2(ix, parent, c);
	ik.x0 = l + 1; ik.x2 = dev_ix_L2(ix, parent, c + 1) - l; ik.x1 = dev_ix_L2(ix, !parent, 3 - c) + 1; ik.info = 0;
}

// push L.ik on the forward list (written downwards from the top of bufA): the entry pushed last is entry 0 of the backward
// sweep's first row and stays in `head`; the one that was there moves to its slot in memory
BSX_HD void seed_fwd_push(SeedLane &L)
{
	if (L.ncurr > 0) seed_list_at(L, L.list_cap - L.ncurr) = L.head;
	L.head = seed_pack(L.ik);
	++L.ncurr;
}

// Run the machine until it needs a bwt_extend (returns 1, request in L.ext_*) or the task is done (0).
// HOT_ONLY: only the states a lane is in on almost every trip (forward walk, backward sweep, LAST-like walk) are compiled in; on
// reaching any other state the function returns 2 and leaves it to the full machine.  The lanes of a wave are in different
// states and the wave executes the code of every state some lane is in, so the rare transitions (pass control, SMEM
// prologue and epilogue) are kept out of the per-trip path.
template <bool HOT_ONLY>
BSX_HD int seed_advance_t(SeedLane &L, const DevIndex &ix, const SeedParams &P)
{
	for (;;) {
		if (HOT_ONLY && L.state != SD_FWD && L.state != SD_BWD_ELEM && L.state != SD_S1 && L.state != SD_FWD_DONE && L.state != SD_P3) return L.state == SD_DONE ? 0 : 2;
		switch (L.state) {
		case SD_DONE: return 0;
		case SD_P1:  // pass 1: SMEMs from every position (memchain.c:65-73)
			if (L.pass_x >= L.len) { L.old_n = L.mem_n; L.k2 = 0; L.state = SD_P2; break; }
			if (seed_qbase(L, L.pass_x) < 4) { L.x0 = L.pass_x; L.min_intv = P.start_width; L.ret_state = SD_P1; L.state = SD_SMEM_BEGIN; }
			else ++L.pass_x;
			break;
		case SD_P2:  // pass 2: re-seed from the middle of long, rare SMEMs (memchain.c:76-85)
			if (L.k2 >= L.old_n) { L.pass_x = 0; L.state = P.max_mem_intv > 0 ? SD_P3 : SD_DONE; break; }
			{
				int kk = L.k2++;
				if (kk < L.mem_cap) {
					DevIntv p = seed_mem_at(L, kk);
					int start = (int)(p.info >> 32), end = (int)(uint32_t)p.info;
					if (end - start < P.split_len || p.x2 > (uint64_t)P.split_width) break;
					L.x0 = (start + end) >> 1; L.min_intv = (int)(p.x2 + 1); L.ret_state = SD_P2; L.state = SD_SMEM_BEGIN;
				}
			}
			break;
		case SD_P3:  // pass 3: LAST-like forward-only seeds (memchain.c:88-103)
			if (L.pass_x >= L.len) { L.state = SD_DONE; break; }
			if (seed_qbase(L, L.pass_x) < 4) {
				L.x0 = L.pass_x;
				seed_set_intv(ix, L.parent, seed_qbase(L, L.x0), L.ik);
				L.i = L.x0 + 1; L.state = SD_S1;
			} else ++L.pass_x;
			break;
		case SD_S1:  // bwt_seed_strategy1 loop (bwt.c:384-394)
			if (L.i >= L.len) { L.pass_x = L.len; L.state = SD_P3; break; }
			{
				int b = seed_qbase(L, L.i);
				if (b < 4) { L.ext_in = L.ik; L.ext_back = 0; L.ext_c = 3 - b; L.ext_which = 1; L.state = SD_S1_POST; return 1; }
				L.pass_x = L.i + 1; L.state = SD_P3;
			}
			break;
		case SD_SMEM_BEGIN: // bwt_smem1a prologue (bwt.c:313-322); max_intv is always 0 here
			L.last_beg = -1;
			if (seed_qbase(L, L.x0) > 3) { L.ret = L.x0 + 1; L.state = SD_SMEM_END; break; }
			if (L.min_intv < 1) L.min_intv = 1;
			seed_set_intv(ix, L.parent, seed_qbase(L, L.x0), L.ik);
			L.ik.info = (uint64_t)(L.x0 + 1);
			L.i = L.x0 + 1; L.ncurr = 0;          // forward list goes into bufA, top-down
			L.state = SD_FWD;
			break;
		case SD_FWD: // forward extension through the complementary index (bwt.c:324-339)
			if (L.i >= L.len) { seed_fwd_push(L); L.state = SD_FWD_DONE; break; }
			{
				int b = seed_qbase(L, L.i);
				if (b < 4) { L.ext_in = L.ik; L.ext_back = 0; L.ext_c = 3 - b; L.ext_which = 1; L.state = SD_FWD_POST; return 1; }
				seed_fwd_push(L); L.state = SD_FWD_DONE;
			}
			break;
		case SD_FWD_DONE: // the list is already "reversed": smallest interval first (bwt.c:341-343)
			L.prev_off = L.list_cap - L.ncurr; L.nprev = L.ncurr;
			L.ret = (int)(L.head.hi >> 8);
			L.i = L.x0 - 1;
			// first backward row set up here (x0 >= 0, so i >= -1): one trip through the switch less per SMEM
			{
				int b = L.i < 0 ? 4 : seed_qbase(L, L.i);
				L.c = b < 4 ? b : -1;
			}
			L.j = 0; L.ncurr = 0; L.have_next = 0;
			L.state = SD_BWD_ELEM;
			break;
		case SD_BWD_ELEM:
			if (L.j >= L.nprev) { // end of the row (bwt.c:362-363): roll over to the next one and go on with its first element
				if (L.ncurr == 0) { L.state = SD_SMEM_END; break; }
				L.prev_off = 0; L.nprev = L.ncurr;
				--L.i;
					if (L.i < -1) { L.state = SD_SMEM_END; break; }
				{
					int b = L.i < 0 ? 4 : seed_qbase(L, L.i);
					L.c = b < 4 ? b : -1;
				}
				L.j = 0; L.ncurr = 0; L.have_next = 0;   // nprev = the survivors of the row just finished: at least one
			}
			{
				{ // j == 0: entry 0 lives in `head` (field by field: a select between the two structs would put them in memory)
					SeedEnt e;
					e.x0 = L.have_next ? L.next_in.x0 : L.head.x0; e.x1 = L.have_next ? L.next_in.x1 : L.head.x1;
					e.x2 = L.have_next ? L.next_in.x2 : L.head.x2; e.hi = L.have_next ? L.next_in.hi : L.head.hi;
					L.ext_in = seed_unpack(e);
				}
				L.have_next = L.j + 1 < L.nprev;
				if (L.have_next) L.next_in = seed_list_at(L, L.prev_off + L.j + 1);
				if (L.c >= 0) { L.ext_back = 1; L.ext_c = L.c; L.ext_which = 0; L.state = SD_BWD_POST; return 1; }
				// c < 0: cannot extend -> candidate SMEM (bwt.c:350-355)
				if (L.ncurr == 0 && (L.last_beg < 0 || L.i + 1 < L.last_beg)) {
					int end = (int)(uint32_t)L.ext_in.info;
					if (end - (L.i + 1) >= P.min_seed_len) seed_emit(L, L.ext_in, L.i + 1, end);
					L.last_beg = L.i + 1;
				}
				++L.j;
			}
			break;
		case SD_SMEM_END: // back to the caller (bwt_smem1's return value only matters to pass 1)
			if (L.ret_state == SD_P1) L.pass_x = L.ret;
			L.state = L.ret_state;
			break;
		default: return 0;
		}
	}
}

BSX_HD int seed_advance(SeedLane &L, const DevIndex &ix, const SeedParams &P) { return seed_advance_t<false>(L, ix, P); }

// Consume the result of the requested extend.
BSX_HD void seed_post(SeedLane &L, const DevIntv &ok, const SeedParams &P)
{
	switch (L.state) {
	case SD_FWD_POST:
		if (ok.x2 != L.ik.x2) { // interval size changed: record the old one (bwt.c:329-333)
			if (L.ncurr < L.list_cap) seed_fwd_push(L); else { L.overflow = 1; ++L.ncurr; }
			if (ok.x2 < (uint64_t)L.min_intv) { L.state = SD_FWD_DONE; break; }
		}
		L.ik = ok; L.ik.info = (uint64_t)(L.i + 1);
		++L.i; L.state = SD_FWD;
		break;
	case SD_BWD_POST:
		if (ok.x2 < (uint64_t)L.min_intv) { // cannot be extended further (bwt.c:350-355)
			if (L.ncurr == 0 && (L.last_beg < 0 || L.i + 1 < L.last_beg)) {
				int end = (int)(uint32_t)L.ext_in.info;
				if (end - (L.i + 1) >= P.min_seed_len) seed_emit(L, L.ext_in, L.i + 1, end);
				L.last_beg = L.i + 1;
			}
		} else { // survives: keep unless it has the size of the previous survivor (bwt.c:357-360)
			if (L.ncurr == 0 || ok.x2 != L.last_x2) {
				DevIntv o = ok; o.info = L.ext_in.info;
				if (L.ncurr == 0) L.head = seed_pack(o); else seed_list_at(L, L.ncurr) = seed_pack(o);
				++L.ncurr;
				L.last_x2 = ok.x2;
			}
		}
		++L.j; L.state = SD_BWD_ELEM;
		break;
	case SD_S1_POST:
		if (ok.x2 < (uint64_t)P.max_mem_intv && L.i - L.x0 >= P.min_seed_len) { // bwt.c:387-391
			if (ok.x2 > 0) seed_emit(L, ok, L.x0, L.i + 1);
			L.pass_x = L.i + 1; L.state = SD_P3;
		} else { L.ik = ok; ++L.i; L.state = SD_S1; }
		break;
	default: break;
	}
}
