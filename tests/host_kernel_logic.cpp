// tests/host_kernel_logic.cpp -- compiles the per-lane state machine of the seed kernel
// (biscuit_amd/csrc/hip/seed_core.hpp, the code k_seed runs per lane) with g++ so its logic can be
// checked on a machine without a GPU.  Test infrastructure only.
#include <vector>
#include <algorithm>
#include <cstring>
extern "C" {
#include "bsx_core.h"
}
#include "seed_core.hpp"

static DevFmi mk(const bsx_fmi_t *f)
{
	DevFmi g; g.primary = f->primary; for (int k = 0; k < 5; ++k) g.L2[k] = f->L2[k];
	g.seq_len = f->seq_len; g.bwt = f->bwt; g.sa = f->sa; g.sa_mask = f->sa_intv - 1; g.sa_shift = 5;
	return g;
}

extern "C" int hostlogic_seed(const bsx_index_t *idx, const bsx_opt_t *opt, const uint8_t *reads, int64_t n, const bsx_seed_task_t *tasks,
                              int mem_cap, uint64_t *out, int64_t out_cap, int64_t *out_off, uint64_t ctr[2])
{
	DevIndex IX; IX.fmi[0] = mk(&idx->fmi[0]); IX.fmi[1] = mk(&idx->fmi[1]); IX.pac = idx->pac; IX.l_pac = idx->ref.l_pac;
	DevFmi *F = IX.fmi;
	SeedParams P;
	P.min_seed_len = opt->min_seed_len; P.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	P.split_width = opt->split_width; P.max_mem_intv = (int)opt->max_mem_intv; P.start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;
	int64_t tot = 0;
	ctr[0] = ctr[1] = 0;
	for (int64_t t = 0; t < n; ++t) {
		int len = tasks[t].len, list_cap = len + 2;
		std::vector<SeedEnt> A(list_cap);
		std::vector<DevIntv> M(mem_cap);
		SeedLane L;
		L.bufA = A.data(); L.mem = M.data(); L.list_cap = list_cap; L.mem_cap = mem_cap; L.stride = 1; L.lane = 0; L.qlds = nullptr;
		L.q = reads + tasks[t].qoff; L.len = len; L.parent = tasks[t].parent;
		seed_lane_begin(L);
		out_off[t] = tot;
		if (len >= P.min_seed_len) {
			while (seed_advance(L, IX, P)) {
				const DevFmi &f = L.ext_which ? F[!L.parent] : F[L.parent];
				DevIntv ok = dev_extend(f, L.ext_in, L.ext_back, L.ext_c, L.n_slow, L.n_fast);
				seed_post(L, ok, P);
			}
			if (L.overflow) return -1;
			std::sort(M.begin(), M.begin() + L.mem_n, [](const DevIntv &a, const DevIntv &b) { return a.info < b.info; });
			if (tot + L.mem_n > out_cap) return -2;
			memcpy(out + tot * 4, M.data(), sizeof(DevIntv) * L.mem_n);
			tot += L.mem_n;
			ctr[0] += 2ull * L.n_slow; ctr[1] += L.n_fast;
		}
	}
	out_off[n] = tot;
	return 0;
}

// dev_fetch_window as the 64 lanes of a wave run it (the region kernels' reference window in LDS)
extern "C" void hostlogic_window(const uint8_t *pac, int64_t l_pac, int64_t beg, int span, uint8_t *win)
{
	for (int lane = 0; lane < 64; ++lane) dev_fetch_window(win, pac, l_pac, beg, span, lane);
}

// ---- the table form of the seeding machine (seed_tab.hpp): the table built with the host's bwt_extend, the machine run lane by lane
#include "seed_tab.hpp"
static void tab_build_host(const DevIndex &IX, int parent, int K, std::vector<SeedEnt> &T)
{
	T.assign(seed_tab_entries(K), SeedEnt{0, 0, 0, 0});
	const DevFmi &o = IX.fmi[!parent];
	for (uint32_t d = 0; d < 3; ++d) { DevIntv ik; seed_set_intv(IX, parent, seed_letter(d, parent), ik); T[d] = seed_pack(ik); }
	uint64_t pw = 3;
	for (int l = 1; l < K; ++l, pw *= 3) {
		const uint64_t at = (pw - 3) >> 1, nx = (pw * 3 - 3) >> 1;
		for (uint64_t key = 0; key < pw; ++key) {
			const DevIntv p = seed_unpack(T[at + key]);
			if (p.x2 == 0) continue;
			uint64_t tk[4], tl[4];
			dev_2occ4(o, p.x1 - 1, p.x1 - 1 + p.x2, tk, tl);
			seed_tab_children(p, tk, tl, o.primary, o.L2, parent, &T[nx + key * 3]);
		}
	}
}
extern "C" int hostlogic_seed_tab(const bsx_index_t *idx, const bsx_opt_t *opt, const uint8_t *reads, int64_t n, const bsx_seed_task_t *tasks, int K,
                                  int mem_cap, uint64_t *out, int64_t out_cap, int64_t *out_off, uint64_t ctr[3])
{
	DevIndex IX; IX.fmi[0] = mk(&idx->fmi[0]); IX.fmi[1] = mk(&idx->fmi[1]); IX.pac = idx->pac; IX.l_pac = idx->ref.l_pac;
	DevFmi *F = IX.fmi;
	SeedParams P;
	P.min_seed_len = opt->min_seed_len; P.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	P.split_width = opt->split_width; P.max_mem_intv = (int)opt->max_mem_intv; P.start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;
	std::vector<SeedEnt> T[2];
	if (K >= 2) { tab_build_host(IX, 0, K, T[0]); tab_build_host(IX, 1, K, T[1]); } else K = 0;
	int64_t tot = 0;
	ctr[0] = ctr[1] = ctr[2] = 0;
	for (int64_t t = 0; t < n; ++t) {
		int len = tasks[t].len, list_cap = len + 2;
		std::vector<SeedEnt> A(list_cap);
		std::vector<DevIntv> M(mem_cap);
		SeedLane2 L;
		memset(&L, 0, sizeof(L));
		L.bufA = A.data(); L.mem = M.data(); L.mstride = sizeof(DevIntv); L.list_cap = list_cap; L.mem_cap = mem_cap; L.stride = 1; L.lane = 0; L.qlds = nullptr;
		L.q = reads + tasks[t].qoff; L.len = len; L.parent = tasks[t].parent;
		std::vector<uint32_t> packed;
		if (len <= 192 && (t & 1) == 0) { // every other strand search reads its bases as the kernel does: two-bit digits, sixteen to a word, words 64 apart
			packed.assign(64 * 12, 0u);
			for (int i = 0; i < len; ++i) packed[(size_t)(i >> 4) << 6] |= seed2_digit_of(L.q[i], L.parent) << ((i & 15) << 1);
			L.qlds = packed.data();
		}
		seed2_lane_begin(L);
		out_off[t] = tot;
		if (len >= P.min_seed_len) {
			int kind;
			while ((kind = seed2_advance(L, IX, P, K)) != SQ_NONE) {
				DevIntv ok;
				if (kind == SQ_FM) {
					const DevFmi &f = L.ext_which ? F[!L.parent] : F[L.parent];
					ok = dev_extend(f, L.ik, L.ext_back, L.ext_c, L.n_slow, L.n_fast);
				} else { ok = seed_unpack(T[L.parent][L.tab_idx]); ok.info = 0; ++L.n_look; }
				seed2_post(L, ok, IX, P, K);
			}
			if (L.overflow) return -1;
			std::sort(M.begin(), M.begin() + L.mem_n, [](const DevIntv &a, const DevIntv &b) { return a.info < b.info; });
			if (tot + L.mem_n > out_cap) return -2;
			memcpy(out + tot * 4, M.data(), sizeof(DevIntv) * L.mem_n);
			tot += L.mem_n;
			ctr[0] += 2ull * L.n_slow; ctr[1] += L.n_fast; ctr[2] += L.n_look;
		}
	}
	out_off[n] = tot;
	return 0;
}
