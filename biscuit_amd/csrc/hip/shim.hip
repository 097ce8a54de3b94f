// shim.hip -- the thin C-ABI shim between the C host code and the HIP kernels (include/bsx.h).
// Owns: device selection, the HBM-resident index (both converted FM indices + SA samples + pac),
// per-chunk read buffer, job/result staging, kernel timing (HIP events on the launch stream) and
// the work counters behind the algorithmic-bytes model.  No CPU fallback: every entry point
// returns BSX_E_NODEVICE when HIP is unusable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <vector>
#include <mutex>
#include "devbuf.hpp"
#include "kernels.h"
#include "index_build.h"
#include "tune.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[bsx-hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return BSX_E_NODEVICE; } } while (0)

// One lane = one HIP stream with its own staging: a chunk is bound to a lane for its whole life, so the front half
// (seeding .. regions) of one chunk and the back half (merge .. SAM) of the previous one can be in flight together.
#define BSX_LANES 6
struct Lane {
	hipStream_t st = nullptr;      // front-half kernels (low priority)
	hipEvent_t ev_seed_done = nullptr, ev_regions_done = nullptr;   // what the next chunk's launches of the same stage wait for
	hipStream_t st_hi = nullptr;   // back-half kernels (K5, K6): high priority, so that they get compute units while another chunk's front half runs
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, ev4 = nullptr, ev5 = nullptr, ev6 = nullptr;   // (ev5, ev6: around the second seeding pass inside the main sequence)
	hipEvent_t tier_ev[12] = {};   // $BSX_PHASES: between the region launches
	hipStream_t st_cp = nullptr;   // unmasked: the front half's small copies (xfer)
	hipEvent_t ev_cp = nullptr;
	hipStream_t st3 = nullptr;     // the last HBM tier of what is known to need it when the occurrences are counted: beside the chunk's other tiers
	hipEvent_t ev_t3a = nullptr, ev_t3b = nullptr;
	hipStream_t st2 = nullptr;     // side stream of the front half: seeding redone with larger lists while the region kernels run
	DevScoring sc;         // set by set_opt on this lane; read by every launch of this lane
	DevBuf reads; size_t n_reads = 0;
	DevBuf gath;             // lists gathered for the download of strand searches the caller takes back
	DevBuf qpack;            // the chunk's reads as base-3 digits for the seeding kernel (k_seedt.hip)
	int64_t rb_tasks = 0;    // strand searches of the last regions batch (their regions, offsets and counts are still in regs / regmeta)
	int flt_key[3] = {-1, -1, -1};   // what fltab was made for
	DevBuf fltab, jobs, res, scratch, scratch2, out, aux, pool, regs, regmeta, slabs, slabs3, slabflags, redo, pos, posoff, xpool, xmeta, x4jobs, tags, mdpool, dd, sswjobs, c2rslab;
	DevBuf small;          // counters[4] | out_cursor | task_cursor | region cursors
	HostBuf hstage;        // pinned staging for bulk results
	HostBuf pin;           // two pinned halves through which large host<->device copies are streamed
	hipEvent_t pev[2] = {nullptr, nullptr};
	// strand searches being seeded again on the side stream while the caller goes on (lane_regions_finish collects them)
	struct {
		bool active = false;
		std::vector<int64_t> tasks;            // their indices in the chunk
		std::vector<bsx_seed_task_t> sub;      // kept alive for the asynchronous upload
		unsigned int n2u = 0;
		HostBuf hres;                          // pinned: region offsets (8 B each) then counts (4 B each)
		hipEvent_t ev = nullptr, ev_tiers = nullptr;
		unsigned long long used_main = 0;      // regions the caller already holds
		unsigned long long regs_cap = 0;       // size of the device's region pool for this chunk
	} rs;
	double k_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int64_t k_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	uint64_t work[5] = {0, 0, 0, 0, 0};   // the region kernels' work since the last reset: strand searches, SA intervals, occurrences, regions, read bases
	// what the last de-duplication of this lane left in `dd` (k_msw.hip reads the reads' lists from there): layout and validity
	struct { bool valid = false; int64_t n_reads = 0; int per_read = 0; size_t o_idx = 0, o_off = 0, o_pool = 0; bool with_long = false; } ddl;
	DevBuf msw_jobs, msw_res, msw_meta, msw_roff; int64_t msw_token = -1;
	std::mutex hi_mu;      // the back half's batches (K5, K6) of this lane, one at a time: the slices of a chunk's back half call them from two threads
	long last_overflow = -1;   // strand searches the first seeding pass of this lane's last chunk left to the second (-1: no chunk yet)
	double seed2_ms = 0; int64_t seed2_launches = 0; uint64_t seed2_tasks = 0;   // the second seeding pass inside the chunk's sequence (its own launch, its own counters: SEED2_CTR)
};
// The seeding launches of a chunk count their FM blocks and table entries in blocks of their own, so that each pass's bytes can be put over
// that pass's time (bsx_device_seed_passes): u64 slots [0], [1], [120] of the lane's counter block for the chunk-wide first pass (and the batch
// form), the same three at SEED2_CTR for the second pass inside the sequence, at SEED3_CTR for the few seeded again on the side stream.
#define SEED2_CTR 256
#define SEED3_CTR 384
#define SMALL_BYTES 4096

struct bsx_device {
	int ordinal = 0;
	char name[256];
	int n_cu = 0;
	DevIndex ix; bool has_index = false;
	DevBuf bwt[2], sa[2], pac, ctg, seedtab[2];
	Lane lane[BSX_LANES];   // scoring matrices and penalties are per lane (Lane::sc): chunks with different options may be in flight together
	// Front halves of consecutive chunks are chained stage by stage (the seeding launch of chunk k+1 waits for that of chunk k, the
	// region launches likewise): four chunks that share the device evenly all finish at the same moment, and the device then idles
	// through the first of their back halves; chained, they are one stage apart and a chunk's seeding overlaps its predecessor's regions
	std::mutex chain_mu;
	hipEvent_t chain_seed = nullptr, chain_regions = nullptr;
};
struct LaneRef { bsx_device *d; int lane; };   // what the backend vtable carries as ctx

#include <sys/time.h>
static double bsx_now_s(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + tv.tv_usec * 1e-6; }
// the device's suffix-array sample: every 2nd rank (the files keep every 32nd, bwtindex.c:328,340): bwt_sa walks one LF step on average.
// Measured at hg38 size, k_occ per chunk: 28 ms at every 4th (12.4 GB per index), 16.5 at every 2nd (24.8 GB), 8.9 with the whole array (49.6 GB)
#define BSX_DEVICE_SA_INTV_DEFAULT 2
static inline unsigned long long *dev_counters(Lane &L) { return (unsigned long long*)L.small.p; }

extern "C" BSX_API int bsx_device_open(int ordinal, bsx_device_t **out)
{
	int n = 0;
	*out = nullptr;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fprintf(stderr, "[bsx-hip] no HIP device available\n"); return BSX_E_NODEVICE; }
	if (ordinal < 0 || ordinal >= n) return BSX_E_ARG;
	HIPCHK(hipSetDevice(ordinal));
	// host threads that wait for the device sleep instead of polling: a chunk stream has half a dozen threads waiting at any time (front
	// halves, K5/K6 batches), and on a 16-core quota their polling was a third of the host's CPU time (the runtime's default polls)
	if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
	bsx_device *d = new bsx_device();
	d->ordinal = ordinal;
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, ordinal));
	snprintf(d->name, sizeof(d->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
	d->n_cu = prop.multiProcessorCount;
	for (int l = 0; l < BSX_LANES; ++l) {
		Lane &L = d->lane[l];
		int lo = 0, hi = 0;
		HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // lo = least, hi = greatest priority (numerically lower)
		// The front-half streams leave a few compute units alone (one in every `reserve`): workgroups of k_seed live for tens of
		// milliseconds and are not preempted, so without free CUs the short high-priority batches of the back half (K5, K6)
		// would queue behind them no matter their priority.
		const int reserve = (int)bsx_tune_long("reserve_cu_every", 0);   // (0 since round 6: with the CUs that are left free spread over every XCD -- below -- the back half's batches start at once and everything else gets slower by more: 735 k reads/s against 776 k without a mask and 757-780 k with the old one)
		bool masked = false;
		if (reserve >= 2 && d->n_cu >= 16) {
			std::vector<uint32_t> mask((size_t)(d->n_cu + 31) / 32, 0u);
			// Which CUs a mask bit stands for (tools/ubench/mask_probe.hip, round 6): bit i is CU i / 32 of shader engine (i / 8) % 4 of XCD i % 8 -- consecutive
			// bits go round the XCDs, then the shader engines.  Workgroups go round the XCDs as well, and a kernel is done when its last workgroup is:
			// the CUs left free must be in EVERY XCD.  Rounds 4-6 cleared every `reserve`-th bit -- with 8, all of XCD 7 and nothing of the others, and
			// a batch of the back half waited a workgroup's lifetime of whatever filled the other seven (the probe: 21 ms against 0.05 ms).  Now the TOP
			// n_cu / reserve bits (whole rows of 32: one CU of every shader engine of every XCD per row) are the ones left out.
			int n_free = d->n_cu / reserve;
			if (d->n_cu >= 64) n_free = (n_free + 31) / 32 * 32;
			if (n_free >= d->n_cu) n_free = d->n_cu / 2;
			for (int cu = 0; cu < d->n_cu - n_free; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
			masked = hipExtStreamCreateWithCUMask(&L.st, (uint32_t)mask.size(), mask.data()) == hipSuccess &&
			         hipExtStreamCreateWithCUMask(&L.st2, (uint32_t)mask.size(), mask.data()) == hipSuccess &&
			         hipExtStreamCreateWithCUMask(&L.st3, (uint32_t)mask.size(), mask.data()) == hipSuccess;
			if (!masked) { (void)hipGetLastError(); if (L.st) { (void)hipStreamDestroy(L.st); L.st = nullptr; } if (L.st2) { (void)hipStreamDestroy(L.st2); L.st2 = nullptr; } if (L.st3) { (void)hipStreamDestroy(L.st3); L.st3 = nullptr; } }
		}
		if (!masked) {
			HIPCHK(hipStreamCreateWithPriority(&L.st, hipStreamNonBlocking, lo));
			HIPCHK(hipStreamCreateWithPriority(&L.st2, hipStreamNonBlocking, lo));
			HIPCHK(hipStreamCreateWithPriority(&L.st3, hipStreamNonBlocking, lo));
		}
		HIPCHK(hipEventCreate(&L.ev_t3a));
		HIPCHK(hipEventCreate(&L.ev_t3b));
		HIPCHK(hipStreamCreateWithPriority(&L.st_hi, hipStreamNonBlocking, hi));
		if (masked && bsx_tune_long("small_copies_unmasked", 1)) { HIPCHK(hipStreamCreateWithPriority(&L.st_cp, hipStreamNonBlocking, hi)); HIPCHK(hipEventCreateWithFlags(&L.ev_cp, hipEventDisableTiming)); }
		HIPCHK(hipEventCreate(&L.ev3));
		HIPCHK(hipEventCreateWithFlags(&L.ev_seed_done, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&L.ev_regions_done, hipEventDisableTiming));
		HIPCHK(hipEventCreate(&L.ev4));
		HIPCHK(hipEventCreate(&L.ev5));
		HIPCHK(hipEventCreate(&L.ev6));
		HIPCHK(hipEventCreateWithFlags(&L.rs.ev, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&L.rs.ev_tiers, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&L.pev[0], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&L.pev[1], hipEventDisableTiming));
		if (L.slabflags.reserve((size_t)d->n_cu * 16 * 4) != BSX_OK) return BSX_E_NOMEM;
		HIPCHK(hipMemset(L.slabflags.p, 0, (size_t)d->n_cu * 16 * 4));
		HIPCHK(hipEventCreate(&L.ev0));
		HIPCHK(hipEventCreate(&L.ev1));
		HIPCHK(hipEventCreate(&L.ev2));
		if (L.small.reserve(SMALL_BYTES) != BSX_OK) return BSX_E_NOMEM;   // u64 slots 32..47: per-stage cycle counts of the region kernels ($BSX_PHASES)
		HIPCHK(hipMemset(L.small.p, 0, SMALL_BYTES));
	}
	memset(&d->ix, 0, sizeof(d->ix));
	for (int l = 0; l < BSX_LANES; ++l) memset(&d->lane[l].sc, 0, sizeof(DevScoring));
	*out = d;
	return BSX_OK;
}

extern "C" BSX_API void bsx_device_close(bsx_device_t *d)
{
	if (!d) return;
	(void)hipSetDevice(d->ordinal);
	devbuf_drain(d->ordinal);   // the blocks this device's buffers left behind when they grew
	for (int i = 0; i < 2; ++i) { d->bwt[i].release(); d->sa[i].release(); d->seedtab[i].release(); }
	d->pac.release(); d->ctg.release();
	for (int l = 0; l < BSX_LANES; ++l) {
		Lane &L = d->lane[l];
		L.reads.release(); L.qpack.release(); L.gath.release(); L.jobs.release(); L.res.release(); L.scratch.release(); L.scratch2.release(); L.out.release(); L.aux.release(); L.pool.release();
		L.small.release(); L.hstage.release(); L.regs.release(); L.regmeta.release(); L.slabs.release(); L.slabflags.release(); L.slabs3.release(); L.redo.release(); L.pin.release(); L.pos.release(); L.posoff.release(); L.xpool.release(); L.xmeta.release(); L.x4jobs.release(); L.fltab.release(); L.flt_key[0] = -1; L.tags.release(); L.mdpool.release(); L.dd.release(); L.c2rslab.release(); L.sswjobs.release(); L.msw_jobs.release(); L.msw_res.release(); L.msw_meta.release(); L.msw_roff.release();
		if (L.pev[0]) (void)hipEventDestroy(L.pev[0]);
		if (L.pev[1]) (void)hipEventDestroy(L.pev[1]);
		if (L.ev0) (void)hipEventDestroy(L.ev0);
		if (L.ev1) (void)hipEventDestroy(L.ev1);
		if (L.ev2) (void)hipEventDestroy(L.ev2);
		if (L.ev_seed_done) (void)hipEventDestroy(L.ev_seed_done);
		if (L.ev_regions_done) (void)hipEventDestroy(L.ev_regions_done);
		if (L.st) (void)hipStreamDestroy(L.st);
		if (L.st_hi) (void)hipStreamDestroy(L.st_hi);
		if (L.st2) (void)hipStreamDestroy(L.st2);
		if (L.st3) (void)hipStreamDestroy(L.st3);
		if (L.st_cp) (void)hipStreamDestroy(L.st_cp);
		if (L.ev_cp) (void)hipEventDestroy(L.ev_cp);
		if (L.ev_t3a) (void)hipEventDestroy(L.ev_t3a);
		if (L.ev_t3b) (void)hipEventDestroy(L.ev_t3b);
		if (L.ev3) (void)hipEventDestroy(L.ev3);
		if (L.ev4) (void)hipEventDestroy(L.ev4);
		if (L.ev5) (void)hipEventDestroy(L.ev5);
		if (L.ev6) (void)hipEventDestroy(L.ev6);
		for (int k = 0; k < 12; ++k) if (L.tier_ev[k]) { (void)hipEventDestroy(L.tier_ev[k]); L.tier_ev[k] = nullptr; }
		if (L.rs.ev) (void)hipEventDestroy(L.rs.ev);
		if (L.rs.ev_tiers) (void)hipEventDestroy(L.rs.ev_tiers);
		L.rs.hres.release();
	}
	delete d;
}

extern "C" BSX_API const char *bsx_device_name(const bsx_device_t *d) { return d ? d->name : ""; }

// pac and the contig table: what the kernels need of the reference besides the FM indices
static int upload_ref(bsx_device_t *d, const bsx_index_t *idx)
{
	size_t npac = (size_t)(idx->ref.l_pac / 4 + 1);
	int rc;
	// ranks and interval sizes travel as 34-bit numbers in the seeding kernel's interval lists (seed_core.hpp: SeedEnt)
	if (idx->ref.l_pac >= (int64_t)1 << 33) { fprintf(stderr, "[bsx-hip] genomes of 2^33 bases (8.6 Gbp) or more are not supported on the device\n"); return BSX_E_ARG; }
	if ((rc = d->pac.reserve(npac + 16)) != BSX_OK) return rc;
	HIPCHK(hipMemcpy(d->pac.p, idx->pac, npac, hipMemcpyHostToDevice));
	d->ix.pac = (const uint8_t*)d->pac.p; d->ix.l_pac = idx->ref.l_pac;
	{ // contig table: offsets (n_seqs + 1, the last one = l_pac) then the is_alt bytes
		const int ns = idx->ref.n_seqs;
		std::vector<int64_t> off((size_t)ns + 1);
		std::vector<uint8_t> alt((size_t)ns + 8, 0);
		for (int i = 0; i < ns; ++i) { off[i] = idx->ref.anns[i].offset; alt[i] = idx->ref.anns[i].is_alt ? 1 : 0; }
		off[ns] = idx->ref.l_pac;
		if ((rc = d->ctg.reserve(((size_t)ns + 1) * 8 + (size_t)ns + 8)) != BSX_OK) return rc;
		HIPCHK(hipMemcpy(d->ctg.p, off.data(), ((size_t)ns + 1) * 8, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy((char*)d->ctg.p + ((size_t)ns + 1) * 8, alt.data(), (size_t)ns, hipMemcpyHostToDevice));
		d->ix.ctg_off = (const int64_t*)d->ctg.p; d->ix.ctg_alt = (const uint8_t*)d->ctg.p + ((size_t)ns + 1) * 8; d->ix.n_seqs = ns;
	}
	return BSX_OK;
}

// The table of k-mer intervals of both converted indices (seed_tab.hpp), built from the indices now resident: K levels, K chosen from the
// text's size (18 for an hg38-sized genome: 2 x 9.3 GB), $BSX_SEED_TAB_K overrides (0: no table, every seeding step an FM extension).
// What the index's optional structures may take of the device's memory: the chunks in flight need room too (their buffers grow with the chunk:
// ~25 GB a lane for 1 M reads against an hg38-sized genome), so a third of the device (at most 64 GB) is left alone.  On a 288 GB part
// nothing changes; on a smaller one the table loses levels and the suffix-array sample gets sparser instead of the upload failing.
static size_t hbm_budget(void)
{
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return ~(size_t)0; }
	const size_t keep = std::min<size_t>(tot / 3, (size_t)64 << 30);
	return fr > keep ? fr - keep : 0;
}
static int build_seed_tables(bsx_device_t *d)
{
	int K = seed_tab_depth(d->ix.fmi[0].seq_len);
	const int K_want = K;
	if (const char *e = bsx_tune_str("seed_tab_k")) { K = atoi(e); if (K < 2) K = 0; if (K > 19) K = 19; }
	d->ix.tab.K = 0; d->ix.tab.t[0] = d->ix.tab.t[1] = nullptr; d->ix.tab.pad_ = 0;
	d->seedtab[0].release(); d->seedtab[1].release();
	while (K >= 2 && 2 * (size_t)seed_tab_entries(K) * sizeof(SeedEnt) > hbm_budget()) --K;   // a level is a third of the table
	Lane &L = d->lane[0];
	for (; K >= 2; --K) { // (and one level fewer when the reservation fails all the same)
		int rc = BSX_OK;
		for (int i = 0; i < 2 && rc == BSX_OK; ++i) rc = d->seedtab[i].reserve_exact((size_t)seed_tab_entries(K) * sizeof(SeedEnt));
		if (rc == BSX_OK) break;
		d->seedtab[0].release(); d->seedtab[1].release();
		(void)hipGetLastError();
	}
	if (K != K_want && !bsx_tune_is_set("seed_tab_k")) fprintf(stderr, "[W::%s] device memory: table of k-mer intervals with %d levels instead of %d%s\n", "bsx-hip", K < 2 ? 0 : K, K_want, K < 2 ? " (none: every seeding step is an FM extension)" : "");
	if (K < 2) { d->seedtab[0].release(); d->seedtab[1].release(); return BSX_OK; }
	for (int i = 0; i < 2; ++i) {
		int rc;
		if ((rc = seedtab_build(L.st, d->ix, i, K, d->seedtab[i].p)) != BSX_OK) return rc;
	}
	HIPCHK(hipStreamSynchronize(L.st));
	HIPCHK(hipGetLastError());
	d->ix.tab.t[0] = (const uint4*)d->seedtab[0].p; d->ix.tab.t[1] = (const uint4*)d->seedtab[1].p; d->ix.tab.K = K;
	return BSX_OK;
}

extern "C" BSX_API int bsx_device_upload_index(bsx_device_t *d, const bsx_index_t *idx)
{
	if (!d || !idx) return BSX_E_ARG;
	if (!idx->fmi[0].bwt || !idx->fmi[1].bwt || !idx->fmi[0].sa || !idx->fmi[1].sa) return BSX_E_ARG;   // no FM indices on the host side: bsx_device_build_index makes them in place
	HIPCHK(hipSetDevice(d->ordinal));
	for (int i = 0; i < 2; ++i) {
		const bsx_fmi_t *f = &idx->fmi[i];
		int rc;
		if ((rc = d->bwt[i].reserve((size_t)f->bwt_size * 4 + 64)) != BSX_OK) return rc;
		if ((rc = d->sa[i].reserve((size_t)f->n_sa * 8)) != BSX_OK) return rc;
		HIPCHK(hipMemset((char*)d->bwt[i].p + (size_t)f->bwt_size * 4, 0, 64));   // the slack the last thread of k_bwt_planes reads and writes
		HIPCHK(hipMemcpy(d->bwt[i].p, f->bwt, (size_t)f->bwt_size * 4, hipMemcpyHostToDevice));
		launch_bwt_planes(d->lane[0].st, (uint32_t*)d->bwt[i].p, (unsigned long long)f->bwt_size);   // the device's block layout (dev_common.hpp); ahead of everything that reads symbols
		HIPCHK(hipStreamSynchronize(d->lane[0].st));
		HIPCHK(hipMemcpy(d->sa[i].p, f->sa, (size_t)f->n_sa * 8, hipMemcpyHostToDevice));
		DevFmi &g = d->ix.fmi[i];
		g.primary = f->primary; for (int k = 0; k < 5; ++k) g.L2[k] = f->L2[k];
		g.seq_len = f->seq_len; g.bwt = (const uint32_t*)d->bwt[i].p; g.sa = (const uint64_t*)d->sa[i].p;
		g.sa_mask = (uint32_t)f->sa_intv - 1; g.sa_shift = 0;
		while ((1 << g.sa_shift) < f->sa_intv) ++g.sa_shift;
		if ((1 << g.sa_shift) != f->sa_intv) return BSX_E_FORMAT;
	}
	int rc;
	if ((rc = upload_ref(d, idx)) != BSX_OK) return rc;
	{ // denser suffix-array sample for the device (the files' 1-in-32 stays what the loader and the host see)
		const char *e = bsx_tune_str("device_sa_intv");
		int want = e ? atoi(e) : BSX_DEVICE_SA_INTV_DEFAULT;
		const int file_intv = (int)d->ix.fmi[0].sa_mask + 1;
		if (want >= 1 && (want & (want - 1)) == 0) { // sparser when the device is short of memory (hbm_budget), down to the files' own sample
			const int asked = want;
			while (want < file_intv && 2 * ((size_t)(d->ix.fmi[0].seq_len / (unsigned)want) + 1) * 8 > hbm_budget()) want <<= 1;
			if (want != asked) fprintf(stderr, "[W::%s] device memory: suffix-array sample every %d ranks instead of every %d\n", "bsx-hip", want < file_intv ? want : file_intv, asked);
		}
		if (want >= 1 && want < file_intv && (want & (want - 1)) == 0 && d->ix.fmi[1].sa_mask == d->ix.fmi[0].sa_mask) {
			DevBuf dense[2];
			Lane &L = d->lane[0];
			for (int i = 0; i < 2; ++i) {
				const unsigned long long nd = d->ix.fmi[i].seq_len / (unsigned)want + 1;
				if ((rc = dense[i].reserve((size_t)nd * 8)) != BSX_OK) { dense[0].release(); dense[1].release(); return rc; }
				launch_sa_dense(L.st, d->n_cu, d->ix, i, (unsigned)want, nd, (unsigned long long*)dense[i].p);
			}
			HIPCHK(hipStreamSynchronize(L.st));
			HIPCHK(hipGetLastError());
			int shift = 0;
			while ((1 << shift) < want) ++shift;
			for (int i = 0; i < 2; ++i) {
				d->sa[i].release();
				d->sa[i] = dense[i];
				d->ix.fmi[i].sa = (const uint64_t*)d->sa[i].p; d->ix.fmi[i].sa_mask = (uint32_t)want - 1; d->ix.fmi[i].sa_shift = (uint32_t)shift;
			}
		}
	}
	if ((rc = build_seed_tables(d)) != BSX_OK) return rc;
	d->has_index = true;
	return BSX_OK;
}

// (f)1 on the device: both FM indices built in HBM from the genome's pac (k_index.hip) and left resident
extern "C" BSX_API int bsx_device_build_index(bsx_device_t *d, bsx_index_t *idx, int fill_host)
{
	if (!d || !idx || !idx->pac || idx->ref.l_pac <= 0) return BSX_E_ARG;
	HIPCHK(hipSetDevice(d->ordinal));
	int rc;
	if ((rc = upload_ref(d, idx)) != BSX_OK) return rc;
	const char *e = bsx_tune_str("device_sa_intv");
	int dense = e ? atoi(e) : BSX_DEVICE_SA_INTV_DEFAULT;
	if (dense < 1 || dense > 32 || (dense & (dense - 1))) dense = BSX_DEVICE_SA_INTV_DEFAULT;
	Lane &L = d->lane[0];
	for (int i = 1; i >= 0; --i) {
		bsx_fmi_t *f = &idx->fmi[i], meta;
		memset(&meta, 0, sizeof(meta));
		free(f->bwt); free(f->sa); f->bwt = nullptr; f->sa = nullptr;
		d->bwt[i].release(); d->sa[i].release();
		const uint64_t n = (uint64_t)idx->ref.l_pac * 2;
		uint32_t *h_bwt = nullptr; uint64_t *h_sa = nullptr;
		if (fill_host) {
			const uint64_t n_occ = (n + 127) / 128 + 1, words = ((n + 15) >> 4) + n_occ * 8, n_sa = (n + 32) / 32;
			h_bwt = (uint32_t*)calloc(words + 16, 4); h_sa = (uint64_t*)calloc(n_sa, 8);
			if (!h_bwt || !h_sa) { free(h_bwt); free(h_sa); return BSX_E_NOMEM; }
		}
		rc = bsx_ix_build_fmi(L.st, d->n_cu, (const uint8_t*)d->pac.p, idx->ref.l_pac, i, dense, 32, &d->bwt[i], &d->sa[i], &meta, h_bwt, h_sa);
		if (rc != BSX_OK) { free(h_bwt); free(h_sa); return rc; }
		launch_bwt_planes(L.st, (uint32_t*)d->bwt[i].p, (unsigned long long)meta.bwt_size);   // (the host copy above is the file layout)
		HIPCHK(hipStreamSynchronize(L.st));
		*f = meta; f->bwt = h_bwt; f->sa = h_sa;   // bwt == NULL: the FM index lives on the device only
		DevFmi &g = d->ix.fmi[i];
		g.primary = meta.primary; for (int k = 0; k < 5; ++k) g.L2[k] = meta.L2[k];
		g.seq_len = meta.seq_len; g.bwt = (const uint32_t*)d->bwt[i].p; g.sa = (const uint64_t*)d->sa[i].p;
		g.sa_mask = (uint32_t)dense - 1; g.sa_shift = 0;
		while ((1 << g.sa_shift) < dense) ++g.sa_shift;
	}
	devbuf_drain(d->ordinal);   // the blocks the builder's growing buffers left behind on this device go now
	if ((rc = build_seed_tables(d)) != BSX_OK) return rc;
	d->has_index = true;
	return BSX_OK;
}

static int lane_set_opt(bsx_device_t *d, int lane, const bsx_opt_t *o)
{
	if (!d || !o) return BSX_E_ARG;
	DevScoring &sc = d->lane[lane].sc;
	memcpy(sc.ctmat, o->ctmat, 25); memcpy(sc.gamat, o->gamat, 25);
	sc.o_del = o->o_del; sc.e_del = o->e_del; sc.o_ins = o->o_ins; sc.e_ins = o->e_ins; sc.zdrop = o->zdrop; sc.a = o->a;
	sc.mx_ct = sc.mx_ga = 0;   // as ksw_extend2 computes it: the maximum starts from 0
	for (int k = 0; k < 25; ++k) { sc.mx_ct = std::max<int>(sc.mx_ct, sc.ctmat[k]); sc.mx_ga = std::max<int>(sc.mx_ga, sc.gamat[k]); }
	return BSX_OK;
}
extern "C" BSX_API int bsx_device_set_opt(bsx_device_t *d, const bsx_opt_t *o) { return lane_set_opt(d, 0, o); }   // the batch calls of include/bsx.h run on lane 0

// Large copies between pageable host memory and the device go through two pinned halves, copy and DMA overlapped:
// handing pageable memory to hipMemcpy makes the runtime pin and unpin the user pages on every call, which costs
// more system time per chunk than the copies themselves.  Synchronous: complete on return.
#define XFER_CHUNK ((size_t)8 << 20)
// The back half's batches (K5, K6: a megabyte of jobs up, a megabyte of results down, on the lane's high-priority stream) do not go through
// the copy engines: those serve every stream first come first served, and behind another chunk's region download (half a gigabyte) a
// one-megabyte copy waited half a second (measured: "download 0.656 s" of a 15 ms K5 batch).  A kernel on the batch's own stream moves the
// bytes between the lane's pinned halves (mapped into the device's address space) and HBM instead.
// the strand searches seeded again inside the main launch sequence: their new lists replace the first pass's in the chunk's offset / count arrays
__global__ void __launch_bounds__(256)
k_patch_lists(const int *which, int n, const long long *off2, const int *cnt2, long long *off, int *cnt)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) { off[which[j]] = off2[j]; cnt[which[j]] = cnt2[j]; }
}
__global__ void __launch_bounds__(256)
k_copy_bytes(const unsigned char *src, unsigned char *dst, size_t n)
{
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
	if ((((size_t)src | (size_t)dst) & 15) == 0) {
		const size_t n16 = n >> 4;
		for (size_t i = tid; i < n16; i += nth) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
		for (size_t i = (n16 << 4) + tid; i < n; i += nth) dst[i] = src[i];
	} else if ((((size_t)src | (size_t)dst) & 3) == 0) {
		const size_t n4 = n >> 2;
		for (size_t i = tid; i < n4; i += nth) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
		for (size_t i = (n4 << 2) + tid; i < n; i += nth) dst[i] = src[i];
	} else for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
}
static void copy_by_kernel(hipStream_t st, void *dst, const void *src, size_t n)
{
	const unsigned int blocks = (unsigned int)std::min<size_t>(std::max<size_t>(1, (n / 16 + 255) / 256), 2048);
	hipLaunchKernelGGL(k_copy_bytes, dim3(blocks), dim3(256), 0, st, (const unsigned char*)src, (unsigned char*)dst, n);
}
static int xfer(Lane &L, hipStream_t st, void *dst, const void *src, size_t n, bool h2d)
{
	if (n == 0) return BSX_OK;
	const bool zc = st == L.st_hi;   // the back half's small batches move with a kernel on their own stream, not through the copy engines (round 4)
	if (!zc && n < ((size_t)256 << 10)) {
		// A small copy is a kernel of the runtime's (__amd_rocclr_copyBuffer: 1 400 of them per chunk in the trace), and on a front-half stream it is
		// dispatched to that stream's CUs only: with other chunks' long-lived workgroups on all of them it waits for one to leave -- tens of milliseconds
		// for eight bytes, at every point where the host reads a count.  It goes to the lane's unmasked stream instead, behind an event of `st`: the
		// CUs the front-half streams leave alone are in every XCD (bsx_device_open), so it runs at once.
		hipStream_t cs = L.st_cp && (st == L.st || st == L.st2 || st == L.st3) ? L.st_cp : st;
		if (cs != st) { HIPCHK(hipEventRecord(L.ev_cp, st)); HIPCHK(hipStreamWaitEvent(cs, L.ev_cp, 0)); }
		HIPCHK(hipMemcpyAsync(dst, src, n, h2d ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, cs));
		HIPCHK(hipStreamSynchronize(cs));
		return BSX_OK;
	}
	int rc;
	if ((rc = L.pin.reserve(2 * XFER_CHUNK)) != BSX_OK) return rc;
	char *pin = (char*)L.pin.p;
	size_t off = 0, prev_off = 0, prev_m = 0;
	int i = 0;
	for (; off < n; off += XFER_CHUNK, ++i) {
		const size_t m = std::min(XFER_CHUNK, n - off);
		char *p = pin + (size_t)(i & 1) * XFER_CHUNK;
		if (h2d) {
			if (i >= 2) HIPCHK(hipEventSynchronize(L.pev[i & 1]));   // the DMA that last read this half is done
			memcpy(p, (const char*)src + off, m);
			if (zc) copy_by_kernel(st, (char*)dst + off, p, m); else HIPCHK(hipMemcpyAsync((char*)dst + off, p, m, hipMemcpyHostToDevice, st));
			HIPCHK(hipEventRecord(L.pev[i & 1], st));
		} else {
			if (zc) copy_by_kernel(st, p, (const char*)src + off, m); else HIPCHK(hipMemcpyAsync(p, (const char*)src + off, m, hipMemcpyDeviceToHost, st));
			HIPCHK(hipEventRecord(L.pev[i & 1], st));
			if (i >= 1) { // drain the previous half while this one is in flight
				HIPCHK(hipEventSynchronize(L.pev[(i - 1) & 1]));
				memcpy((char*)dst + prev_off, pin + (size_t)((i - 1) & 1) * XFER_CHUNK, prev_m);
			}
			prev_off = off; prev_m = m;
		}
	}
	HIPCHK(hipStreamSynchronize(st));
	if (!h2d) memcpy((char*)dst + prev_off, pin + (size_t)((i - 1) & 1) * XFER_CHUNK, prev_m);
	return BSX_OK;
}
#define H2D(st, dst, src, n) do { int rc_ = xfer(L, st, dst, src, n, true); if (rc_ != BSX_OK) return rc_; } while (0)
#define D2H(st, dst, src, n) do { int rc_ = xfer(L, st, dst, src, n, false); if (rc_ != BSX_OK) return rc_; } while (0)

static int lane_set_reads(bsx_device_t *d, int lane, const uint8_t *buf, size_t n)
{
	if (!d) return BSX_E_ARG;
	Lane &L = d->lane[lane];
	HIPCHK(hipSetDevice(d->ordinal));
	int rc;
	if ((rc = L.reads.reserve(n + 64)) != BSX_OK) return rc;
	H2D(L.st, L.reads.p, buf, n);
	L.n_reads = n;
	return BSX_OK;
}

extern "C" BSX_API int bsx_device_set_reads(bsx_device_t *d, const uint8_t *buf, size_t n) { return lane_set_reads(d, 0, buf, n); }

// time one kernel (already enqueued between ev0/ev1 on the lane's stream)
static int finish_timed(Lane &L, int k)
{
	float ms = 0;
	HIPCHK(hipEventSynchronize(L.ev1));
	HIPCHK(hipEventElapsedTime(&ms, L.ev0, L.ev1));
	L.k_ms[k] += ms; L.k_launch[k] += 1;
	HIPCHK(hipGetLastError());
	return BSX_OK;
}

extern "C" BSX_API int bsx_device_counters(bsx_device_t *d, uint64_t c[4], int reset)
{
	if (!d) return BSX_E_ARG;
	HIPCHK(hipSetDevice(d->ordinal));
	c[0] = c[1] = c[2] = c[3] = 0;
	for (int l = 0; l < BSX_LANES; ++l) {
		uint64_t t[4];
		HIPCHK(hipMemcpy(t, d->lane[l].small.p, 32, hipMemcpyDeviceToHost));
		for (int k = 0; k < 4; ++k) c[k] += t[k];
		if (reset) HIPCHK(hipMemset(d->lane[l].small.p, 0, 32));
		for (int b = 0; b < 2; ++b) { // the later seeding passes' FM blocks (counted apart: bsx_device_seed_passes)
			char *p = (char*)d->lane[l].small.p + (size_t)(b ? SEED3_CTR : SEED2_CTR) * 8;
			HIPCHK(hipMemcpy(t, p, 16, hipMemcpyDeviceToHost));
			c[0] += t[0]; c[1] += t[1];
			if (reset) HIPCHK(hipMemset(p, 0, 16));
		}
	}
	return BSX_OK;
}

// table entries read by the seeding kernel since the last reset (summed over the lanes), and the depth of the resident table
extern "C" BSX_API int bsx_device_seed_table(bsx_device_t *d, uint64_t *lookups, int *depth, int reset)
{
	if (!d) return BSX_E_ARG;
	HIPCHK(hipSetDevice(d->ordinal));
	uint64_t tot = 0;
	for (int l = 0; l < BSX_LANES; ++l) {
		static const int at[3] = {120, SEED2_CTR + 120, SEED3_CTR + 120};
		for (int b = 0; b < 3; ++b) {
			uint64_t t = 0;
			HIPCHK(hipMemcpy(&t, (char*)d->lane[l].small.p + (size_t)at[b] * 8, 8, hipMemcpyDeviceToHost));
			tot += t;
			if (reset) HIPCHK(hipMemset((char*)d->lane[l].small.p + (size_t)at[b] * 8, 0, 8));
		}
	}
	if (lookups) *lookups = tot;
	if (depth) *depth = d->has_index ? d->ix.tab.K : 0;
	return BSX_OK;
}

// The seeding passes of bsx_regions_batch one by one since the last reset (call it BEFORE bsx_device_counters / _seed_table with reset, which
// zero the same blocks): w[0], w[1] FM blocks and table entries read by the chunk-wide first pass (and by bsx_seed_batch launches), w[2], w[3]
// by the second pass inside the chunk's sequence, w[4] its launches, w[5] its strand searches; ms[0], ms[1]: their summed HIP-event times.
extern "C" BSX_API int bsx_device_seed_passes(bsx_device_t *d, uint64_t w[6], double ms[2], int reset)
{
	if (!d || !w || !ms) return BSX_E_ARG;
	HIPCHK(hipSetDevice(d->ordinal));
	for (int k = 0; k < 6; ++k) w[k] = 0;
	ms[0] = ms[1] = 0;
	for (int l = 0; l < BSX_LANES; ++l) {
		Lane &L = d->lane[l];
		uint64_t t[2], u = 0;
		HIPCHK(hipMemcpy(t, L.small.p, 16, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(&u, (char*)L.small.p + 120 * 8, 8, hipMemcpyDeviceToHost));
		w[0] += t[0] + t[1]; w[1] += u;
		HIPCHK(hipMemcpy(t, (char*)L.small.p + SEED2_CTR * 8, 16, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(&u, (char*)L.small.p + (SEED2_CTR + 120) * 8, 8, hipMemcpyDeviceToHost));
		w[2] += t[0] + t[1]; w[3] += u;
		w[4] += (uint64_t)L.seed2_launches; w[5] += L.seed2_tasks;
		ms[0] += L.k_ms[0]; ms[1] += L.seed2_ms;
		if (reset) { L.seed2_ms = 0; L.seed2_launches = 0; L.seed2_tasks = 0; }
	}
	return BSX_OK;
}

extern "C" BSX_API int bsx_device_region_work(bsx_device_t *d, uint64_t w[5], int reset)
{
	if (!d || !w) return BSX_E_ARG;
	for (int k = 0; k < 5; ++k) w[k] = 0;
	for (int l = 0; l < BSX_LANES; ++l) for (int k = 0; k < 5; ++k) { w[k] += d->lane[l].work[k]; if (reset) d->lane[l].work[k] = 0; }
	return BSX_OK;
}

extern "C" BSX_API int bsx_device_kernel_time(bsx_device_t *d, int k, double *total_ms, int64_t *launches, int reset)
{
	if (!d || k < 0 || k >= 8) return BSX_E_ARG;
	double ms = 0; int64_t n = 0;
	for (int l = 0; l < BSX_LANES; ++l) {
		ms += d->lane[l].k_ms[k]; n += d->lane[l].k_launch[k];
		if (reset) { d->lane[l].k_ms[k] = 0; d->lane[l].k_launch[k] = 0; }
	}
	if (total_ms) *total_ms = ms;
	if (launches) *launches = n;
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K1+K2
// ------------------------------------------------------------------------------------------
static bool intv_info_lt(const bsx_intv_t &a, const bsx_intv_t &b) { return a.info < b.info; }

struct SeedAsm {
	const long long *off; const int *cnt; const bsx_intv_t *dense;   // round-0 results (dense, arbitrary task order)
	const int64_t *redo_of; const std::vector<std::vector<bsx_intv_t>> *redo;
	bsx_intv_t *out; const int64_t *out_off;
};
static void seed_asm_worker(void *data, long i, int tid)
{
	const SeedAsm *A = (const SeedAsm*)data;
	(void)tid;
	bsx_intv_t *dst = A->out + A->out_off[i];
	const int64_t n = A->out_off[i + 1] - A->out_off[i];
	if (n <= 0) return;
	if (A->redo_of && A->redo_of[i] >= 0) std::copy((*A->redo)[A->redo_of[i]].begin(), (*A->redo)[A->redo_of[i]].end(), dst);
	else memcpy(dst, A->dense + A->off[i], sizeof(bsx_intv_t) * (size_t)n);
	// ks_introsort(mem_intv) (memchain.c:105): records with equal info are identical, any sort gives the reference order
	if (n > 1) std::sort(dst, dst + n, intv_info_lt);
}

static int lane_seed_batch(bsx_device_t *d, int lane, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
                           bsx_intv_t **out, int64_t *out_cap, int64_t *out_off)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) { out_off[0] = 0; return BSX_OK; }
	HIPCHK(hipSetDevice(d->ordinal));
	int rc, max_len = 0;
	for (int64_t i = 0; i < n; ++i) max_len = std::max(max_len, tasks[i].len);
	if (max_len >= 1 << 18) return BSX_E_ARG;   // a lane addresses its slab with 32-bit byte offsets (seed_core.hpp): 64 lanes x 32 B x (8 x max_len) entries of the second pass must stay below 2^32
	SeedParams P;
	P.min_seed_len = opt->min_seed_len;
	P.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	P.split_width = opt->split_width;
	P.max_mem_intv = (int)opt->max_mem_intv;
	P.start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;

	std::vector<long long> h_off((size_t)n);
	std::vector<int> h_n((size_t)n);
	std::vector<int64_t> todo;           // tasks whose interval list did not fit: redone with more room
	std::vector<bsx_seed_task_t> sub;
	int mem_cap = std::max(64, max_len);
	unsigned long long dense_cap = (unsigned long long)n * 24 * (unsigned long long)((max_len + 149) / 150 > 1 ? (max_len + 149) / 150 : 1) + 4096;   // (in proportion to the read length)
	std::vector<std::vector<bsx_intv_t>> redo_results;
	std::vector<int64_t> redo_index;
	const bsx_intv_t *h_dense = nullptr;
	std::vector<bsx_intv_t> dense_sub;

	for (int round = 0; round < 6; ++round) {
		const bsx_seed_task_t *cur = tasks; int64_t cn = n;
		if (round > 0) {
			if (todo.empty()) break;
			sub.resize(todo.size());
			for (size_t i = 0; i < todo.size(); ++i) sub[i] = tasks[todo[i]];
			cur = sub.data(); cn = (int64_t)sub.size();
			mem_cap *= 8; dense_cap = (unsigned long long)cn * mem_cap + 4096;
		}
		int list_cap = max_len + 2;
		int waves = (int)std::min<int64_t>((cn + 63) / 64, (int64_t)d->n_cu * 16);
		const size_t per_lane = (size_t)mem_cap * 32 + (size_t)list_cap * 16;   // mem_cap SMEMs of 32 bytes, one list of list_cap 16-byte entries
		waves = (int)std::max<size_t>(4, std::min<size_t>((size_t)waves, ((size_t)8 << 30) / (64 * per_lane)));   // (long reads seeded again with long lists: fewer waves, not tens of GB)
		int grid = (waves + 3) / 4;
		size_t lanes = (size_t)grid * 256;
		size_t scratch_bytes = lanes * per_lane;
		if ((rc = L.scratch.reserve(scratch_bytes)) != BSX_OK) return rc;
		if ((rc = L.jobs.reserve((size_t)cn * sizeof(bsx_seed_task_t))) != BSX_OK) return rc;
		if ((rc = L.out.reserve((size_t)dense_cap * sizeof(DevIntv))) != BSX_OK) return rc;
		if ((rc = L.aux.reserve((size_t)cn * 12 + 64)) != BSX_OK) return rc;
		if ((rc = L.qpack.reserve(seedt_pack_bytes(cn))) != BSX_OK) return rc;
		long long *d_off = (long long*)L.aux.p; int *d_n = (int*)((char*)L.aux.p + (size_t)cn * 8);
		unsigned long long *ctr = dev_counters(L);
		HIPCHK(hipMemcpyAsync(L.jobs.p, cur, (size_t)cn * sizeof(bsx_seed_task_t), hipMemcpyHostToDevice, L.st));
		HIPCHK(hipMemsetAsync(ctr + 4, 0, 16, L.st));   // out_cursor (u64) + task_cursor (u32)
		HIPCHK(hipEventRecord(L.ev0, L.st));
		launch_seed(L.st, grid, d->ix, (const uint8_t*)L.reads.p, (const bsx_seed_task_t*)L.jobs.p, (int)cn, P,
		            (DevIntv*)L.scratch.p, list_cap, mem_cap, (DevIntv*)L.out.p, dense_cap, ctr + 4, d_off, d_n,
		            (unsigned int*)(ctr + 5), ctr, 0, (unsigned int*)L.slabflags.p, grid * 4, 0, 0, (uint32_t*)L.qpack.p);
		HIPCHK(hipEventRecord(L.ev1, L.st));
		if ((rc = finish_timed(L, 7)) != BSX_OK) return rc;   // the batch form, for what the host chains: not the chunk-wide launch of slot 0
		std::vector<long long> r_off((size_t)cn); std::vector<int> r_n((size_t)cn);
		unsigned long long used = 0;
		D2H(L.st, r_off.data(), d_off, (size_t)cn * 8);
		D2H(L.st, r_n.data(), d_n, (size_t)cn * 4);
		D2H(L.st, &used, ctr + 4, 8);
		if (used > dense_cap) used = dense_cap;
		std::vector<int64_t> next;
		if (round == 0) {
			if ((rc = L.hstage.reserve((size_t)used * sizeof(bsx_intv_t) + 64)) != BSX_OK) return rc;
			if (used) D2H(L.st, L.hstage.p, L.out.p, (size_t)used * sizeof(bsx_intv_t));
			h_dense = (const bsx_intv_t*)L.hstage.p;
			for (int64_t i = 0; i < n; ++i) { h_off[i] = r_off[i]; h_n[i] = r_n[i]; if (r_n[i] < 0) next.push_back(i); }
		} else {
			dense_sub.resize((size_t)used);
			if (used) D2H(L.st, dense_sub.data(), L.out.p, (size_t)used * sizeof(bsx_intv_t));
			for (int64_t i = 0; i < cn; ++i) {
				if (r_n[i] < 0) { next.push_back(todo[i]); continue; }
				redo_index.push_back(todo[i]);
				redo_results.emplace_back(dense_sub.begin() + r_off[i], dense_sub.begin() + r_off[i] + r_n[i]);
				h_n[todo[i]] = r_n[i];
			}
		}
		todo.swap(next);
		if (todo.empty()) break;
	}
	if (!todo.empty()) { fprintf(stderr, "[bsx-hip] seed_batch: %zu tasks still overflow\n", todo.size()); return BSX_E_INTERNAL; }

	// CSR in task order, each list ordered by info; assembled by the host worker pool
	int64_t tot = 0;
	for (int64_t i = 0; i < n; ++i) { out_off[i] = tot; tot += h_n[i]; }
	out_off[n] = tot;
	if (*out_cap < tot) { *out_cap = tot + (tot >> 2) + 16; *out = (bsx_intv_t*)realloc(*out, sizeof(bsx_intv_t) * (size_t)*out_cap); }
	std::vector<int64_t> redo_of;
	if (!redo_index.empty()) { redo_of.assign((size_t)n, -1); for (size_t r = 0; r < redo_index.size(); ++r) redo_of[redo_index[r]] = (int64_t)r; }
	SeedAsm A;
	A.off = h_off.data(); A.cnt = h_n.data(); A.dense = h_dense; A.redo_of = redo_of.empty() ? nullptr : redo_of.data(); A.redo = &redo_results;
	A.out = *out; A.out_off = out_off;
	bsx_parallel_for(bsx_host_threads(opt), seed_asm_worker, &A, (long)n);
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K1+K2 -> K3 + chaining + chain filter + chain-to-region on the device; the interval lists never leave HBM
// ------------------------------------------------------------------------------------------
static int lane_regions_batch(bsx_device_t *d, int lane, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks,
                              bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n,
                              bsx_intv_t **decl_intv, int64_t *decl_cap, int64_t *decl_off)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) return BSX_OK;
	if (n > 0x7fffffff) return BSX_E_ARG;
	HIPCHK(hipSetDevice(d->ordinal));
	struct timespec ts_in, ts_out;
	clock_gettime(CLOCK_MONOTONIC, &ts_in);
	int rc, max_len = 0;
	for (int64_t i = 0; i < n; ++i) max_len = std::max(max_len, tasks[i].len);
	if (max_len >= 1 << 18) return BSX_E_ARG;   // a lane addresses its slab with 32-bit byte offsets (seed_core.hpp): 64 lanes x 32 B x (8 x max_len) entries of the second pass must stay below 2^32
	SeedParams P;
	P.min_seed_len = opt->min_seed_len;
	P.split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	P.split_width = opt->split_width;
	P.max_mem_intv = (int)opt->max_mem_intv;
	P.start_width = (opt->flag & BSX_F_SELF_OVLP) ? 2 : 1;
	RegParams R;
	R.a = opt->a; R.w = opt->w; R.o_del = opt->o_del; R.e_del = opt->e_del; R.o_ins = opt->o_ins; R.e_ins = opt->e_ins;
	R.pen_clip5 = opt->pen_clip5; R.pen_clip3 = opt->pen_clip3; R.min_seed_len = opt->min_seed_len; R.min_chain_weight = opt->min_chain_weight;
	R.max_chain_gap = opt->max_chain_gap; R.max_occ = opt->max_occ; R.bsstrand = opt->bsstrand; R.max_chain_extend = (uint32_t)opt->max_chain_extend;
	R.mask_level = opt->mask_level; R.drop_ratio = opt->drop_ratio; R.prof = bsx_phases() ? 1 : 0;
	R.gap_cap = -1;
	R.walk_on = 1;
	R.ext_win = (int)bsx_tune_long("ext_win", 1);
	// mem_flt_chained_seeds (memchain.c:537-548) by read length: does the seed-SW filter run, and with which threshold.  Tabulated
	// here because the rule goes through log() and the float / double conversions of the reference's expression.
	bool any_flt = false;
	{
		std::vector<int32_t> ft((size_t)max_len + 1, INT32_MIN);
		for (int l = 1; l <= max_len; ++l) {
			const double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log((double)l);
			if (min_l > 0.05f * l) continue;
			ft[l] = (int32_t)(opt->a * min_l + .499);
		}
		for (int64_t i = 0; i < n && !any_flt; ++i) any_flt = tasks[i].len >= 0 && ft[tasks[i].len] != INT32_MIN;   // (for a read of this chunk, not for some length below its longest)
		// (the table follows from the longest read, -W and -A: uploaded again only when one of them changes)
		if (!(L.fltab.p && L.flt_key[0] == max_len && L.flt_key[1] == opt->min_chain_weight && L.flt_key[2] == opt->a)) {
			if ((rc = L.fltab.reserve(ft.size() * 4)) != BSX_OK) return rc;
			H2D(L.st, L.fltab.p, ft.data(), ft.size() * 4);
			L.flt_key[0] = max_len; L.flt_key[1] = opt->min_chain_weight; L.flt_key[2] = opt->a;
		}
		R.flt_tab = (const int32_t*)L.fltab.p; R.flt_len = max_len;
	}
	// Chunks with reads above the short kernels' 256 bases (up to regions_long_max_query(); longer ones are chained by the caller) or with
	// the seed-SW filter active take the instantiations with longer tables, every tier exports its chains, and the filter (k_seedsw)
	// runs ahead of chains -> regions.
	const int long_reads = max_len > 256 ? 1 : 0;
	const bool export_all = long_reads || any_flt;

	// $BSX_SEED_MEM_CAP (tests): a short first-pass list, so that ordinary reads take the seeded-again path too
	const int mem_cap = bsx_tune_is_set("seed_mem_cap") ? std::max(4, (int)bsx_tune_long("seed_mem_cap", 64)) : std::max(64, max_len), list_cap = max_len + 2;
	// room for the interval lists (32 B each) and regions (56 B each) of the whole chunk; repeat-rich genomes average
	// dozens of intervals per strand search, and HBM is not the scarce resource here
	const unsigned long long lf = (unsigned long long)std::max(1, (max_len + 149) / 150);   // pools are sized per 150 bases of read
	// the interval lists: strand search t's own stretch of mem_cap entries (k_seedt writes them where they stay), then room for the lists of the
	// strand searches seeded again with longer lists, which go one behind the other from the cursor
	const bool seed_direct = bsx_tune_long("seed_direct", 1) != 0 && !bsx_tune_is_set("seed_form")
	                         && (unsigned long long)n * (unsigned long long)mem_cap <= (768ull << 20);   // (24 GB of lists: a chunk of short reads with one very long one keeps the lists one behind the other)   // ($BSX_SEED_DIRECT=0: one list behind the other, copied there when a strand search is done)
	const unsigned long long direct_n = seed_direct ? (unsigned long long)n * (unsigned long long)mem_cap : 0;
	const unsigned long long dense_cap = direct_n + (unsigned long long)n * (seed_direct ? 16 : 96) * lf + (1u << 20), regs_cap = (unsigned long long)n * 24 + 65536;   // (a read inside a repeat family has dozens of regions: 6 per strand search overflowed on an hg38-like genome)
	// workgroups with a bounded life (a few tasks per lane / wave), many more of them than fit on the chip
	const int seed_quota = (int)bsx_tune_long("seed_quota", bsx_tune_is_set("seed_form") ? 1 : 0);   // strand searches per lane, 0 = lanes take them until none is left.  The table form (k_seedt.hip) runs persistent lanes: a lane that is done takes the next strand search in the same trip (measured at hg38 scale: 68 ms against 97 with one per lane and 86 with four); the kernel without the table does best with one (292 ms against 335 with two and 359 persistent: its lanes then move through the seeding passes together)
	// extensions after which the first seeding pass hands a strand search to the second one (0: never)
	// (4096 for reads of 150 bases, which need ~1.2 k; in proportion for longer ones)
	const int trip_budget = std::max(0, (int)bsx_tune_long("seed_trip_budget", 4096));   // (the kernel scales it per 256 bases of read)
	const int reg_quota = std::max(1, (int)bsx_tune_long("regions_quota", 16));
	const int n_slabs = d->n_cu * 16;
	const int seed_wpc = 16;   // quota 0 = persistent waves, sixteen per CU
	int grid = seed_quota > 0 ? (int)((n + 256LL * seed_quota - 1) / (256LL * seed_quota))
	                          : (int)((std::min<int64_t>((n + 63) / 64, (int64_t)d->n_cu * seed_wpc) + 3) / 4);
	const size_t lanes = (size_t)n_slabs * 64, scratch_bytes = lanes * ((size_t)mem_cap * 32 + (size_t)list_cap * 16);   // as in lane_seed_batch
	// the HBM tiers: workgroups of four waves over per-wave slabs; they are bound by the latency of their slabs, so what counts is waves in flight:
	// three workgroups per CU for the first (its kernel is held to 168 VGPRs for that and spills: 710 -> 578 ms per chunk on the hg38-like genome
	// all the same), one per CU for the second (1.3 MB of slab a wave; 222 -> 167 ms: a launch lasts as long as its largest strand search)
	// (the last tier: one workgroup of four waves per CU, or two -- "tier3_wgs"; its list longest strand search first -- "tier3_order")
	const int big_grid = d->n_cu * 3, huge_grid = d->n_cu * std::max(1, std::min(3, (int)bsx_tune_long("tier3_wgs", 2)));
	const bool order3 = bsx_tune_long("tier3_order", 1) != 0;
	if ((rc = L.scratch.reserve(scratch_bytes)) != BSX_OK) return rc;
	if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_seed_task_t))) != BSX_OK) return rc;
	if ((rc = L.out.reserve((size_t)dense_cap * sizeof(DevIntv))) != BSX_OK) return rc;
	if ((rc = L.aux.reserve((size_t)n * 12 + 64)) != BSX_OK) return rc;
	if ((rc = L.qpack.reserve(seedt_pack_bytes(n))) != BSX_OK) return rc;
	if ((rc = L.regs.reserve((size_t)regs_cap * sizeof(bsx_region_t))) != BSX_OK) return rc;
	if ((rc = L.regmeta.reserve((size_t)n * 49 + 64)) != BSX_OK) return rc;
	const int c2rh_grid = d->n_cu * 5;   // workgroups of the chains -> regions launch with its tables in HBM (a slab each)
	if ((rc = L.c2rslab.reserve((size_t)c2rh_grid * c2r_hbm_slab_bytes())) != BSX_OK) return rc;
	if ((rc = L.slabs.reserve((size_t)big_grid * 6 * regions_slab_bytes(2))) != BSX_OK) return rc;   // (the exporting form runs three workgroups per CU)
	if ((rc = L.slabs3.reserve((size_t)huge_grid * 4 * regions_slab_bytes(3))) != BSX_OK) return rc;
	// one u64 per seed occurrence of the chunk: ~125 per strand search against an hg38-sized genome (a 3-letter 19-mer has random
	// copies there), a few dozen against small ones.  $BSX_POS_CAP (tests): a cap small enough for strand searches to find no room.
	const unsigned long long pos_cap = bsx_tune_is_set("pos_cap") ? strtoull(bsx_tune_str("pos_cap"), 0, 10) : (unsigned long long)n * 384 * lf + (1u << 20);
	if ((rc = L.pos.reserve((size_t)pos_cap * 8)) != BSX_OK) return rc;
	if ((rc = L.posoff.reserve((size_t)n * 8 + 64)) != BSX_OK) return rc;
	// what the LDS tiers export for the chains -> regions launch: ~0.5 KB per strand search (a task that finds no room goes to the next tier)
	const unsigned long long xcap = (unsigned long long)n * 2560 * lf + (64u << 20);   // (a dozen chains per strand search at hg38 size: 24 + 16 bytes each, 48 more for its extensions)
	if ((rc = L.xpool.reserve((size_t)xcap)) != BSX_OK) return rc;
	if ((rc = L.xmeta.reserve((size_t)n * 12 + 64)) != BSX_OK) return rc;
	// the extensions of the exported chains' best seeds are made ahead of chains -> regions, four to a wavefront (k_ext4.hip); $BSX_X4=0: all
	// of them inline in k_c2r (the tests compare the two).  Not for chunks whose lists the seed-SW filter rewrites after the export.
	const int use_x4 = (int)bsx_tune_long("x4", 1);
	const unsigned long long x4_cap = (unsigned long long)n * 12 * lf + (1u << 20);
	if (use_x4 && !export_all && (rc = L.x4jobs.reserve((size_t)x4_cap * x4_job_bytes())) != BSX_OK) return rc;
	// the seed filter's alignments (reads of 700 bases and more: memchain.c:501-535) are one batch per chunk: a few dozen per strand search
	// ($BSX_SSW_CAP, tests: a list too short for the chunk -- what finds no room in it is aligned a wavefront at a time)
	const unsigned long long ssw_cap = !any_flt ? 0 : bsx_tune_is_set("ssw_cap") ? std::max(8ull, strtoull(bsx_tune_str("ssw_cap"), 0, 10))
	                                   : std::min<unsigned long long>((unsigned long long)n * 24 * lf + (1u << 20), 0x3ffffff0ull);
	if (any_flt && (rc = L.sswjobs.reserve((size_t)ssw_cap * seedsw_job_bytes())) != BSX_OK) return rc;
	unsigned long long *d_pos = (unsigned long long*)L.pos.p; long long *d_posoff = (long long*)L.posoff.p;
	long long *d_off = (long long*)L.aux.p; int *d_n = (int*)((char*)L.aux.p + (size_t)n * 8);
	long long *r_off = (long long*)L.regmeta.p; int *r_n = (int*)((char*)L.regmeta.p + (size_t)n * 8);
	int *retry_a = (int*)((char*)L.regmeta.p + (size_t)n * 12), *retry_b = (int*)((char*)L.regmeta.p + (size_t)n * 16);
	int *retry_m = (int*)((char*)L.regmeta.p + (size_t)n * 20);
	int *retry_l = (int*)((char*)L.regmeta.p + (size_t)n * 24);
	unsigned char *d_cls = (unsigned char*)L.regmeta.p + (size_t)n * 28;
	int *retry_c = (int*)((char*)L.regmeta.p + (((size_t)n * 29 + 3) & ~(size_t)3));   // what the first chains -> regions launch declines (ordinary chunks)
	int *retry_e = retry_c + 4 * n;   // what launch_occ lists for the last HBM tier ahead of everything else (u32 count and cursor: slot 22)
	int *retry_h = retry_c + n, *xlist_t2 = retry_c + 2 * n, *retry_f = retry_c + 3 * n;   // round 6: what the second declines (-> the one with tables in HBM); the strand searches the first HBM tier exports; what takes that tier's full form
	// counters (u64 slots of L.small): [4] interval cursor  [5] seed task cursor  [6] region cursor
	// u32 view from slot 7: [0] tier-1 task cursor [1] tier-2 count [2] tier-2 cursor [3] tier-3 count [4] tier-3 cursor
	//                       [5] redo count [6] redo tier-3 cursor [7] redo seed task cursor  ([8],[9] = u64 slot 11: the K3 cursor)
	//                       [10] count of what the LDS tier in between hands to tier 2  [11] that tier's cursor
	//                       u64 slot 13: cursor of the export pool; slot 14 as two u32: exported task count, k_c2r's cursor
	unsigned long long *ctr = dev_counters(L);
	unsigned int *c32 = (unsigned int*)(ctr + 7);
	RgXPoolArg XA;
	XA.base = (unsigned char*)L.xpool.p; XA.cap = xcap; XA.cursor = ctr + 13; XA.xoff = (long long*)L.xmeta.p; XA.xlist = (int*)((char*)L.xmeta.p + (size_t)n * 8);
	XA.xcount = (unsigned int*)(ctr + 14);
	XA.ext = use_x4 && !export_all ? 1 : 0;
	const uint8_t *d_reads = (const uint8_t*)L.reads.p;
	const bsx_seed_task_t *d_tasks = (const bsx_seed_task_t*)L.jobs.p;
	L.rb_tasks = n;
	L.ddl.valid = false;
	H2D(L.st, L.jobs.p, tasks, (size_t)n * sizeof(bsx_seed_task_t));
	HIPCHK(hipMemsetAsync(ctr + 4, 0, 96, L.st));
	HIPCHK(hipMemsetAsync(ctr + 20, 0, 8, L.st));   // (count and cursor of the second chains -> regions launch)
	HIPCHK(hipMemsetAsync(ctr + 22, 0, 8, L.st));   // (count and cursor of the early launch of the last HBM tier)
	HIPCHK(hipMemsetAsync(ctr + 119, 0, 8, L.st));  // (the first seeding pass's count of strand searches to be seeded again)
	HIPCHK(hipMemsetAsync(ctr + 70, 0, 80, L.st));  // (round 6: counts and cursors of the launches between the second chains -> regions launch and the last HBM tier)
	HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(ctr + 4), (int)(uint32_t)direct_n, 1, L.st));            // the cursor starts behind the strand searches' own stretches
	HIPCHK(hipMemsetD32Async((hipDeviceptr_t)((uint32_t*)(ctr + 4) + 1), (int)(uint32_t)(direct_n >> 32), 1, L.st));
	const int chain = (int)bsx_tune_long("chain_stages", 3);   // 0: none, 1: seeding, 2: seeding and regions, 3: the same but the HBM tiers (a few long strand searches on a few waves) hold nobody back (measured A/B on one box, 16 chunks: 1.76 / 1.81 M reads/s with 2, 2.00 / 1.89 M with 3)
	if (chain >= 1) {
		std::lock_guard<std::mutex> g(d->chain_mu);
		if (d->chain_seed && d->chain_seed != L.ev_seed_done) HIPCHK(hipStreamWaitEvent(L.st, d->chain_seed, 0));
		// 4: front halves one after the other -- a chunk's seeding also waits for the region kernels of the chunk before it
		if (chain >= 4 && d->chain_regions && d->chain_regions != L.ev_regions_done) HIPCHK(hipStreamWaitEvent(L.st, d->chain_regions, 0));
	}
	HIPCHK(hipEventRecord(L.ev0, L.st));
	launch_seed(L.st, grid, d->ix, d_reads, d_tasks, (int)n, P,
	            (DevIntv*)L.scratch.p, list_cap, mem_cap, (DevIntv*)L.out.p, dense_cap, ctr + 4, d_off, d_n,
	            (unsigned int*)(ctr + 5), ctr, seed_quota, (unsigned int*)L.slabflags.p, n_slabs, trip_budget, R.prof, (uint32_t*)L.qpack.p, seed_direct ? 0ull : ~0ull);
	HIPCHK(hipEventRecord(L.ev1, L.st));
	if (chain >= 1) {
		std::lock_guard<std::mutex> g(d->chain_mu);
		HIPCHK(hipEventRecord(L.ev_seed_done, L.st));
		d->chain_seed = L.ev_seed_done;
	}
	// Strand searches whose interval list overflowed are seeded again with lists eight times as long.  A few of them (reads inside tandem
	// repeats: ~300 k dependent FM steps on one lane, tens of milliseconds) go to the side stream and through a tier sequence of their own
	// while the main one runs (below).  MANY of them -- an hg38-like genome: 4 % of the strand searches, reads inside young copies of a repeat
	// family, 23 ms for 43 k -- are seeded again right here and rejoin the chunk's one tier sequence (round 5): the second sequence could only
	// start when the first had ended (shared slabs and export lists), and the front half waited for it as long again as for the first
	// whenever the chunks in flight were in step (the command line's steady state: 8.5 s front halves).  The price is a host round trip
	// between seeding and the suffix-array lookups for every chunk (the counts, 8 MB).
	std::vector<int> first_n;   // the first pass's counts (negative: overflowed)
	std::vector<int> merged_which; const int *merged_cnt = nullptr;   // the strand searches of the second pass inside this sequence, and where their new counts are
	bool merged = false;
	const int budget2_mul = std::max(1, (int)bsx_tune_long("seed_budget2", 8));   // the second pass's budget, in first-pass budgets
	const long merge_min = bsx_tune_long("redo_merge_min", 4096);   // (tests: 1 = always merged, a huge number or a negative one = never)
	// Whether there are that many is the kernel's own count (counters[119]: 8 bytes back, not every strand search's count), and the host waits
	// for it only when the lane's last chunk came anywhere near the threshold: on a clean genome, where a few dozen overflow, nothing stands
	// between a chunk's seeding and its suffix-array lookups after the lane's first chunk.
	unsigned long long n_over = 0;
	const bool may_merge = merge_min >= 0 && merge_min <= (long)n;
	if (may_merge && (L.last_overflow < 0 || L.last_overflow >= merge_min / 4)) {
		HIPCHK(hipEventSynchronize(L.ev1));
		D2H(L.st, &n_over, ctr + 119, 8);
		L.last_overflow = (long)n_over;
	}
	if (may_merge && (long)n_over >= merge_min) {
		first_n.resize((size_t)n);
		D2H(L.st, first_n.data(), d_n, (size_t)n * 4);
		std::vector<int> which;
		for (int64_t i = 0; i < n; ++i) if (first_n[i] < 0) which.push_back((int)i);
		const size_t n2 = which.size();
		const int g2 = (int)((n2 + 255) / 256);
		const long long cap2 = std::max<long long>((long long)mem_cap * 8, 1024);
		const size_t scratch2 = (size_t)g2 * 256 * ((size_t)cap2 * 32 + (size_t)list_cap * 16);
		if ((long)n2 >= merge_min && n2 > 0 && n2 <= 262144 && g2 * 4 <= n_slabs && scratch2 <= ((size_t)24 << 30)) {
			if ((rc = L.scratch2.reserve(scratch2)) != BSX_OK) return rc;
			if ((rc = L.redo.reserve(n2 * (sizeof(bsx_seed_task_t) + 8 + 4 + 4) + 1024)) != BSX_OK) return rc;
			L.rs.sub.resize(n2);
			for (size_t j = 0; j < n2; ++j) L.rs.sub[j] = tasks[which[j]];
			bsx_seed_task_t *t2 = (bsx_seed_task_t*)L.redo.p;
			long long *off2 = (long long*)(t2 + n2); int *cnt2 = (int*)(off2 + n2); int *which_d = cnt2 + n2;
			H2D(L.st, t2, L.rs.sub.data(), n2 * sizeof(bsx_seed_task_t));
			H2D(L.st, which_d, which.data(), n2 * sizeof(int));
			HIPCHK(hipMemsetAsync(ctr + 99, 0, 8, L.st));   // (u32 [7] of the second sequence's cursors: its seed task cursor)
			HIPCHK(hipEventRecord(L.ev5, L.st));
			// (with a budget of its own, eight times the first pass's: the launch lasts as long as its slowest lane, and the handful of reads
			// inside tandem repeats -- hundreds of thousands of dependent FM steps -- made it 90-170 ms for 23 ms of work; they go on to the
			// side stream below like the few of a clean genome)
			launch_seed(L.st, g2, d->ix, d_reads, t2, (int)n2, P, (DevIntv*)L.scratch2.p, list_cap, (int)cap2, (DevIntv*)L.out.p, dense_cap, ctr + 4,
			            off2, cnt2, (unsigned int*)(ctr + 96) + 7, ctr + SEED2_CTR, 0, (unsigned int*)L.slabflags.p, g2 * 4, trip_budget * budget2_mul, bsx_phases() ? 2 : 0, (uint32_t*)L.qpack.p);
			hipLaunchKernelGGL(k_patch_lists, dim3((unsigned int)((n2 + 255) / 256)), dim3(256), 0, L.st, (const int*)which_d, (int)n2, (const long long*)off2, (const int*)cnt2, d_off, d_n);
			HIPCHK(hipEventRecord(L.ev6, L.st));
			merged = true; merged_which.swap(which); merged_cnt = cnt2;
			if (bsx_phases()) fprintf(stderr, "[M::regions_batch] %zu strand searches seeded again inside the main sequence\n", n2);
		}
	}
	if (!merged) { HIPCHK(hipEventRecord(L.ev5, L.st)); HIPCHK(hipEventRecord(L.ev6, L.st)); }
	// The last HBM tier beside the others (round 6): its launch lasts as long as its longest strand search -- 190 ms for a read inside a tandem repeat,
	// on a device it leaves almost empty -- and what it takes is known as soon as the occurrences are counted (more intervals or occurrences than the
	// tier before it holds).  launch_occ lists those, the first tier skips them, and the launch goes to a stream of its own right behind the
	// suffix-array lookups; what the first HBM tier hands on later (a strand search that grew along the way) gets the launch at the end as before.
	// Not with the exporting tiers (kilobase reads, the seed filter), and not under phases=2 (the stage counters are read tier by tier).
	const bool early3 = bsx_tune_long("tier3_early", 1) != 0 && !export_all && !long_reads && bsx_phases() != 2 && L.st3;
	launch_occ(L.st, d->n_cu, d->ix, d_tasks, (int)n, (const DevIntv*)L.out.p, d_off, d_n, opt->max_occ, d_pos, pos_cap, ctr + 11, d_posoff, ctr, d_cls, nullptr,
	           early3 ? retry_e : nullptr, early3 ? (unsigned int*)(ctr + 22) : nullptr);
	HIPCHK(hipEventRecord(L.ev4, L.st));
	if (early3) {
		HIPCHK(hipStreamWaitEvent(L.st3, L.ev4, 0));
		HIPCHK(hipEventRecord(L.ev_t3a, L.st3));
		if (order3) launch_order_list(L.st3, retry_e, (unsigned int*)(ctr + 22), (const DevIntv*)L.out.p, d_off, d_n, opt->max_occ);
		launch_regions_slab(L.st3, 3, huge_grid, d->ix, L.sc, R, d_reads, d_tasks, (const DevIntv*)L.out.p, d_off, d_n,
		                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, r_off, r_n, retry_e, (unsigned int*)(ctr + 22), (unsigned int*)(ctr + 22) + 1, L.slabs3.p, nullptr, nullptr, ctr, d_posoff, d_pos);
		HIPCHK(hipEventRecord(L.ev_t3b, L.st3));
	}
	if (chain >= 2) {
		std::lock_guard<std::mutex> g(d->chain_mu);
		if (d->chain_regions && d->chain_regions != L.ev_regions_done) HIPCHK(hipStreamWaitEvent(L.st, d->chain_regions, 0));
	}
	// tier 1 -> retry_a -> LDS tier with larger tables -> retry_m -> tier 2 (HBM slabs) -> retry_b -> tier 3
	const int use_mid = (int)bsx_tune_long("regions_mid", 1);
	// strand searches a wave of the larger LDS tier / of the chains -> regions launch takes before it leaves (bounded workgroup life)
	const int mid_quota = std::max(1, (int)bsx_tune_long("mid_quota", 8));
	const int c2r_quota = std::max(1, (int)bsx_tune_long("c2r_quota", 16));
	// The tier sequence over a task list (the chunk's, and once more the re-seeded strand searches' on the side stream).  k32: u32 cursors and
	// counts ([0] tier-1 cursor [1] tier-2 count [2] tier-2 cursor [3] tier-3 count [4] tier-3 cursor [5] k_seedsw's cursor [10] what the
	// larger LDS tier hands on [11] its cursor); xc32: exported count | k_c2r's cursor.
	const bool trace_tiers = bsx_phases() != 0 || bsx_tune_long("tiers", 0) != 0;   // $BSX_TIERS: the launch times alone (no cycle counters in the kernels)
	int n_marks = 0; const char *mark_name[12];
	auto run_tiers = [&](hipStream_t st, const bsx_seed_task_t *T, int64_t nT, const long long *offs, const int *cnts, long long *roffs, int *rns,
	                     int *ra, int *rm, int *rb, unsigned int *k32, unsigned int *xc32, const RgXPoolArg &XP, const long long *posoffs,
	                     const unsigned char *clsx, bool main_seq, int *rl, unsigned int *l_count, unsigned int *l_cursor, unsigned int *x4c, int *rc_list, unsigned int *rc32, unsigned int *ssw32,
	                     int *rh_list, int *x2_list, int *rf_list, unsigned int *h32) -> int {
		int rc2;
		const int rgrid = (int)((nT + 4LL * reg_quota - 1) / (4LL * reg_quota));
		// $BSX_PHASES: the main sequence's launches one by one (events between them)
		// $BSX_PHASES=2: the stage counters read (and zeroed) after every launch of the main sequence: where each tier's wave cycles go
		const bool per_tier = bsx_phases() == 2;
#define TIER_MARK(name_) do { if (main_seq && trace_tiers && n_marks < 12) { if (!L.tier_ev[n_marks]) HIPCHK(hipEventCreate(&L.tier_ev[n_marks])); HIPCHK(hipEventRecord(L.tier_ev[n_marks], st)); mark_name[n_marks++] = name_; \
		if (per_tier) { unsigned long long pf_[16]; HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipMemcpy(pf_, ctr + 32, sizeof(pf_), hipMemcpyDeviceToHost)); HIPCHK(hipMemset(ctr + 32, 0, sizeof(pf_))); \
			{ unsigned long long ds_[12]; HIPCHK(hipMemcpy(ds_, ctr + 160, sizeof(ds_), hipMemcpyDeviceToHost)); HIPCHK(hipMemset(ctr + 160, 0, sizeof(ds_))); \
			  if (ds_[2] | ds_[3] | ds_[4] | ds_[6] | ds_[8] | ds_[10]) fprintf(stderr, "[M::regions_batch] %s hands on: %llu for their intervals, %llu for their occurrences (or a list too long), %llu chains, %llu tied chain starts, %llu regions, %llu an interval to be walked further\n", name_, ds_[8], ds_[2], ds_[3], ds_[4], ds_[6], ds_[10]); } \
			if (pf_[11]) fprintf(stderr, "[M::regions_batch] %s: seed loops: %llu seeds reached, %llu skipped as contained, %llu took the extension made ahead, %llu extended in place (%llu extensions, %llu rows)\n", name_, pf_[11], pf_[12], pf_[13], pf_[14], pf_[8], pf_[9]); \
			double tot_ = 0; for (int k_ = 0; k_ < 8; ++k_) tot_ += (double)pf_[k_]; \
			if (tot_ > 0) fprintf(stderr, "[M::regions_batch] %s: intervals %.1f%% occurrences %.1f%% chaining %.1f%% weights+order %.1f%% sort %.1f%% filter %.1f%% prologues+seed tests %.1f%% extension %.1f%% of %.0f M wave cycles\n", name_, \
			                      100 * pf_[0] / tot_, 100 * pf_[1] / tot_, 100 * pf_[2] / tot_, 100 * pf_[3] / tot_, 100 * pf_[4] / tot_, 100 * pf_[5] / tot_, 100 * pf_[6] / tot_, 100 * pf_[7] / tot_, tot_ * 1e-6); } } } while (0)
		TIER_MARK("start");
		launch_regions(st, rgrid, d->ix, L.sc, R, d_reads, T, (int)nT, (const DevIntv*)L.out.p, offs, cnts,
		               (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, k32 + 0, ra, k32 + 1, reg_quota, ctr, posoffs, d_pos, clsx, XP, long_reads);
		if (main_seq) HIPCHK(hipEventRecord(L.ev3, st));
		TIER_MARK("tier 1");
		if (use_mid)
			launch_regions_mid(st, (int)((nT + 2LL * mid_quota - 1) / (2LL * mid_quota)), d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                   (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, ra, k32 + 1, k32 + 11, rm, k32 + 10, ctr, posoffs, d_pos, XP, mid_quota, long_reads);
		TIER_MARK("tier 1b");
		int *to2 = use_mid ? rm : ra; unsigned int *n2c = use_mid ? k32 + 10 : k32 + 1;
		if (use_mid && long_reads) { // kilobase reads: a second LDS tier with larger tables (two workgroups per CU) for what outgrows the first (three)
			launch_regions_mid(st, (int)((nT + 2LL * mid_quota - 1) / (2LL * mid_quota)), d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                   (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rm, k32 + 10, l_cursor, rl, l_count, ctr, posoffs, d_pos, XP, mid_quota, 3);
			to2 = rl; n2c = l_count;
			TIER_MARK("tier 1c");
		}
		const bool use_1c = bsx_tune_long("tier1c", 1) != 0;   // ($BSX_TIER1C=0: the tier sequence of rounds 2-4, for the A/B)
		const bool tier1c = use_mid && !long_reads && !export_all && use_1c;
		if (tier1c) { // ordinary reads inside repeat families: an LDS tier with twice the tables behind the first two (round 5)
			launch_regions_mid(st, (int)((nT + 2LL * mid_quota - 1) / (2LL * mid_quota)), d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                   (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rm, k32 + 10, l_cursor, rl, l_count, ctr, posoffs, d_pos, XP, mid_quota, 4);
			to2 = rl; n2c = l_count;
			TIER_MARK("tier 1c");
		}
		const int c2r_grid = (int)((nT + 4LL * c2r_quota - 1) / (4LL * c2r_quota));
		if (export_all) {
			// every tier exports; then the seed-SW filter where it applies, then chains -> regions (what outgrows its tables is left to the caller)
			launch_regions_slab(st, 2, big_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, to2, n2c, k32 + 2, L.slabs.p, rb, k32 + 3, ctr, posoffs, d_pos, &XP);
			TIER_MARK("tier 2 (exports)");
			launch_regions_slab(st, 3, huge_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rb, k32 + 3, k32 + 4, L.slabs3.p, nullptr, nullptr, ctr, posoffs, d_pos, &XP);
			TIER_MARK("tier 3 (exports)");
			if (any_flt) { // (the side stream's run comes after the main one's on the device -- it waits for the tiers -- and may use the same job list)
				// ssw32: [0] job count [1] the third launch's cursor; k32[5]: the first launch's
				launch_seedsw(st, (int)std::min<int64_t>(nT, (int64_t)d->n_cu * 32), d->n_cu, d->ix, L.sc, R, d_reads, T, XP, k32 + 5, ssw32, L.sswjobs.p, (unsigned int)ssw_cap, ctr);
			}
			TIER_MARK("seed filter");
#ifdef BSX_DEBUG_XCHECK
			if (main_seq) { // debug builds (tools/dbg/lds_variants.sh): every exported record looked at before k_c2r reads it
				unsigned long long used = 0; unsigned int xn = 0;
				HIPCHK(hipStreamSynchronize(st));
				D2H(st, &used, XP.cursor, 8); D2H(st, &xn, XP.xcount, 4);
				HIPCHK(hipStreamSynchronize(st));
				fprintf(stderr, "[xcheck] %u records, %llu bytes of %llu\n", xn, used, XP.cap);
				if (used > XP.cap) used = XP.cap;
				std::vector<unsigned char> a((size_t)used);
				std::vector<long long> xo((size_t)nT); std::vector<int> xl((size_t)xn);
				D2H(st, a.data(), XP.base, (size_t)used); D2H(st, xo.data(), XP.xoff, (size_t)nT * 8); D2H(st, xl.data(), XP.xlist, (size_t)xn * 4);
				HIPCHK(hipStreamSynchronize(st));
				std::vector<std::pair<long long, long long>> span;
				long n_bad = 0;
				for (unsigned int k = 0; k < xn; ++k) {
					const int t = xl[k];
					if (t < 0 || t >= nT) { fprintf(stderr, "[xcheck] list entry %u names task %d\n", k, t); ++n_bad; continue; }
					const long long o = xo[t];
					if (o < 0 || (unsigned long long)o + 24 > used) { fprintf(stderr, "[xcheck] task %d offset %lld\n", t, o); ++n_bad; continue; }
					const int *H = (const int*)(a.data() + o);
					const int nk = H[0], nsd = H[1], has = H[4], tier = H[5];
					const long long bytes = 24 + (long long)nk * 24 + (long long)nsd * 16 + (has ? (long long)nk * 48 : 0);
					span.push_back(std::make_pair(o, o + bytes));
					bool bad = nk <= 0 || nsd < nk || (unsigned long long)(o + bytes) > used;
					long long so_expect = 0;
					for (int c = 0; c < nk && !bad; ++c) {
						const unsigned char *xc = a.data() + o + 24 + (size_t)c * 24;
						const long long pos = *(const long long*)xc; const int rid = *(const int*)(xc + 8), so = *(const int*)(xc + 12);
						const int nm = *(const unsigned short*)(xc + 16), ne = *(const unsigned short*)(xc + 18);
						if (rid < 0 || rid >= d->ix.n_seqs || pos < 0 || pos >= 2 * d->ix.l_pac || so < so_expect || so + nm + ne > nsd) {
							fprintf(stderr, "[xcheck] task %d (len %d, tier tables %d) record at %lld: %d chains %d seeds; chain %d: pos %lld rid %d seed_off %d (expected %lld) main %d extra %d\n",
							        t, T == d_tasks ? tasks[t].len : -1, tier, o, nk, nsd, c, pos, rid, so, so_expect, nm, ne);
							bad = true;
						}
						so_expect = so + nm + ne;   // (k_seedsw shortens a main list in place and moves the backup list up behind it: offsets stay those of the export)
					}
					if (bad) { ++n_bad; if (nk <= 0 || nsd < nk) fprintf(stderr, "[xcheck] task %d (tier tables %d) record at %lld: %d chains %d seeds %lld bytes\n", t, tier, o, nk, nsd, bytes); }
				}
				std::sort(span.begin(), span.end());
				for (size_t k = 1; k < span.size(); ++k) if (span[k].first < span[k - 1].second) { fprintf(stderr, "[xcheck] records overlap: [%lld, %lld) and [%lld, %lld)\n", span[k - 1].first, span[k - 1].second, span[k].first, span[k].second); ++n_bad; }
				fprintf(stderr, "[xcheck] %ld bad\n", n_bad);
			}
#endif
			// (what outgrows the launch's tables -- 64 regions of a strand search, 256 seeds of a list -- gets a second one with up to 1024 of each, the
			// regions in HBM; round 6: those were forty strand searches per chunk of kilobase reads chained on the host, a second of its time each chunk)
			launch_c2r(st, c2r_grid, d->ix, L.sc, R, d_reads, T, XP, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, xc32 + 1, rh_list, h32 + 0, ctr, c2r_quota, long_reads);
			{
				RgXPoolArg XB3 = XP;
				XB3.xlist = rh_list; XB3.xcount = h32 + 0;
				launch_c2r(st, d->n_cu * 2, d->ix, L.sc, R, d_reads, T, XB3, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, h32 + 1, nullptr, nullptr, ctr, 1 << 30, 4, L.c2rslab.p);
			}
			TIER_MARK("chains -> regions");
			return BSX_OK;
		}
		// chains -> regions of everything the two LDS tiers exported; what outgrows its tables joins the list of the HBM tiers.
		// (The first HBM tier exporting its chains as well -- 156 instead of 225 VGPRs, its chains through k_c2r with everybody else's -- was
		// measured in rounds 3 and 4 and removed: what k_c2r cannot hold, 64 regions and 128 seeds a chain, takes the full form afterwards anyway,
		// 1468 against 1287 ms per chunk on the hg38-like genome.  Round 6 takes it up again with a chains -> regions launch that CAN hold them:)
		//
		// h32 (u32): [0] what the second chains -> regions launch declines [1] the third's cursor | [4] what takes the HBM tier's full form [5] its
		// cursor | [6] the exporting launch's cursor
		const int t2x = tier1c && XP.ext ? (int)bsx_tune_long("tier2_export", 0) : 0;   // 1: the first HBM tier in steps (below) instead of its monolithic form (chains, filter and extensions inline in one launch)
		if (XP.ext) {
			launch_x4(st, d->n_cu, d->ix, L.sc, R, d_reads, T, (long long)nT, XP, L.x4jobs.p, x4_cap, x4c, R.prof ? ctr + 56 : nullptr);
			TIER_MARK("extensions");
		}
		if (tier1c) { // ... and what the chains -> regions launch declines (more than 64 regions, lists of more than 128 seeds) gets a second one with larger tables
			launch_c2r(st, c2r_grid, d->ix, L.sc, R, d_reads, T, XP, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, xc32 + 1, rc_list, rc32, ctr, c2r_quota);
			RgXPoolArg XB2 = XP;
			XB2.xlist = rc_list; XB2.xcount = rc32;
			if (t2x) { // ... and what THAT declines (up to 1024 regions, 1024 seeds a list) a third, the regions it makes in HBM: the record and its extensions are there
				launch_c2r(st, d->n_cu * 2, d->ix, L.sc, R, d_reads, T, XB2, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rc32 + 1, rh_list, h32 + 0, ctr, 1 << 30, 2);
				RgXPoolArg XB3 = XP;
				XB3.xlist = rh_list; XB3.xcount = h32 + 0;
				launch_c2r(st, c2rh_grid, d->ix, L.sc, R, d_reads, T, XB3, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, h32 + 1, rf_list, h32 + 4, ctr, 1 << 30, 3, L.c2rslab.p);
			} else
			launch_c2r(st, d->n_cu * 2, d->ix, L.sc, R, d_reads, T, XB2, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rc32 + 1, to2, n2c, ctr, 1 << 30, 2);   // (few strand searches, long ones: persistent waves)
		} else
		launch_c2r(st, c2r_grid, d->ix, L.sc, R, d_reads, T, XP, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, xc32 + 1, to2, n2c, ctr, c2r_quota);
		if (main_seq && chain == 3) { // the HBM tiers (a few long strand searches on a few waves) do not hold the next chunk's region launches back
			std::lock_guard<std::mutex> g(d->chain_mu);
			HIPCHK(hipEventRecord(L.ev_regions_done, st));
			d->chain_regions = L.ev_regions_done;
		}
		TIER_MARK("chains -> regions");
		if (t2x) {
			// The first HBM tier in steps (round 6, "tier2_export=1"; measured and not the default).  What outgrew the LDS tiers'
			// tables is chained and filtered over the tier's slabs and EXPORTED like the LDS tiers' strand searches; its chains' best seeds are extended
			// ahead, several jobs to a wavefront (k_extl / k_ext4) -- inline, a wavefront each, the extensions are 58 % of the tier's cycles --; the seed
			// loop is run by the chains -> regions launch that keeps up to 1024 regions in HBM.  Only what even that cannot hold takes the tier's
			// monolithic form; what outgrows the tier's tables goes on to the last tier as before.
			RgXPoolArg XT = XP;
			XT.xlist = x2_list; XT.xcount = h32 + 2;
			if (t2x >= 2) XT.ext = 2;   // every seed of every main list extended ahead (a slot per seed): the seed loops of these strand searches skip one seed in fourteen
			launch_regions_slab(st, 2, big_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, to2, n2c, h32 + 6, L.slabs.p, rb, k32 + 3, ctr, posoffs, d_pos, &XT);
			TIER_MARK("tier 2 (chains)");
			launch_x4(st, d->n_cu, d->ix, L.sc, R, d_reads, T, (long long)nT, XT, L.x4jobs.p, x4_cap, h32 + 8, nullptr);
			TIER_MARK("tier 2 (extensions)");
			launch_c2r(st, c2rh_grid, d->ix, L.sc, R, d_reads, T, XT, (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, h32 + 3, rf_list, h32 + 4, ctr, 1 << 30, 3, L.c2rslab.p);
			TIER_MARK("tier 2 (chains -> regions)");
			launch_regions_slab(st, 2, big_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
			                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rf_list, h32 + 4, h32 + 5, L.slabs.p, rb, k32 + 3, ctr, posoffs, d_pos);
		} else
		launch_regions_slab(st, 2, big_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
		                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, to2, n2c, k32 + 2, L.slabs.p, rb, k32 + 3, ctr, posoffs, d_pos);
		TIER_MARK("tier 2");
		if (main_seq && early3) HIPCHK(hipStreamWaitEvent(st, L.ev_t3b, 0));   // (the early launch: it shares the tier's slabs, and everything behind this point waits for its regions)
		if (order3) launch_order_list(st, rb, k32 + 3, (const DevIntv*)L.out.p, offs, cnts, opt->max_occ);
		launch_regions_slab(st, 3, huge_grid, d->ix, L.sc, R, d_reads, T, (const DevIntv*)L.out.p, offs, cnts,
		                    (bsx_region_t*)L.regs.p, regs_cap, ctr + 6, roffs, rns, rb, k32 + 3, k32 + 4, L.slabs3.p, nullptr, nullptr, ctr, posoffs, d_pos);
		TIER_MARK("tier 3");
		return BSX_OK;
	};
	if ((rc = run_tiers(L.st, d_tasks, n, d_off, d_n, r_off, r_n, retry_a, retry_m, retry_b, c32, (unsigned int*)(ctr + 14), XA, d_posoff, d_cls, true, retry_l, c32 + 6, c32 + 7, (unsigned int*)(ctr + 16), retry_c, (unsigned int*)(ctr + 20), (unsigned int*)(ctr + 15),
	                    retry_h, xlist_t2, retry_f, (unsigned int*)(ctr + 70))) != BSX_OK) return rc;

	if (chain == 2 || chain >= 4) {
		std::lock_guard<std::mutex> g(d->chain_mu);
		HIPCHK(hipEventRecord(L.ev_regions_done, L.st));
		d->chain_regions = L.ev_regions_done;
	}
	HIPCHK(hipEventRecord(L.rs.ev_tiers, L.st));
	// While those run: strand searches whose interval list overflowed (reads inside tandem repeats: ~300 k dependent FM steps
	// on one lane) are seeded again on the side stream with much longer lists and go through the third tier as well.  None of
	// that is waited for here: everything is enqueued, and lane_regions_finish collects the result when the caller gets to the
	// chunk's back half.
	const bool trace = bsx_phases() != 0;
	struct timespec ts0, ts1, ts2, ts3;
	clock_gettime(CLOCK_MONOTONIC, &ts0);
	std::vector<int64_t> redo;            // task indices
	{
		std::vector<int> s_n;
		if (!first_n.empty()) s_n.swap(first_n);
		else { s_n.resize((size_t)n); HIPCHK(hipEventSynchronize(L.ev1)); D2H(L.st2, s_n.data(), d_n, (size_t)n * 4); }
		for (int64_t i = 0; i < n; ++i) if (s_n[i] < 0) redo.push_back(i); else L.work[1] += (uint64_t)s_n[i];
		L.last_overflow = (long)redo.size();
		if (merged) { // they are in the main sequence, but for those that the second pass gave up on as well (its budget, its list)
			std::vector<int> c2(merged_which.size());
			HIPCHK(hipEventSynchronize(L.ev6));
			D2H(L.st2, c2.data(), merged_cnt, c2.size() * 4);
			redo.clear();
			for (size_t j = 0; j < c2.size(); ++j) if (c2[j] < 0) redo.push_back(merged_which[j]); else L.work[1] += (uint64_t)c2[j];
		}
		if (redo.size() > 262144) redo.clear();   // (one slab per four waves of the second pass: n_slabs bounds it) leave them to the caller
	}
	clock_gettime(CLOCK_MONOTONIC, &ts1);
	L.rs.active = false;
	if (!redo.empty()) {
		// Seeded again on the side stream with lists eight times as long and no trip budget, then through the same tiers as everything else
		// (on a genome with hg38's repeat content these are 4 % of the strand searches -- reads inside young copies of a repeat family -- not
		// the few dozen tandem-repeat reads of a clean one)
		const size_t n2 = redo.size();
		const int g2 = (int)((n2 + 255) / 256);
		const long long cap2 = std::max<long long>((long long)mem_cap * 8, 1024);
		const size_t scratch2 = (size_t)g2 * 256 * ((size_t)cap2 * 32 + (size_t)list_cap * 16);
		if (g2 * 4 <= n_slabs && scratch2 <= ((size_t)24 << 30)) {
			if ((rc = L.scratch2.reserve(scratch2)) != BSX_OK) return rc;
			L.rs.tasks = redo; L.rs.sub.resize(n2); L.rs.n2u = (unsigned int)n2;
			for (size_t j = 0; j < n2; ++j) L.rs.sub[j] = tasks[redo[j]];
			// device side, per re-seeded strand search: task | interval offset | region offset | position offset | export offset | interval
			// count | region count | three tier lists | export list | tier class
			if ((rc = L.redo.reserve(n2 * (sizeof(bsx_seed_task_t) + 8 + 8 + 8 + 8 + 4 + 4 + 16 + 4 + 4 + 12 + 1) + 1024)) != BSX_OK) return rc;
			if ((rc = L.rs.hres.reserve(n2 * 12 + 64)) != BSX_OK) return rc;
			bsx_seed_task_t *t2 = (bsx_seed_task_t*)L.redo.p;
			long long *off2 = (long long*)(t2 + n2); long long *roff2 = off2 + n2; long long *posoff2 = roff2 + n2; long long *xoff2 = posoff2 + n2;
			int *cnt2 = (int*)(xoff2 + n2); int *rn2 = cnt2 + n2; int *ra2 = rn2 + n2, *rm2 = ra2 + n2, *rb2 = rm2 + n2, *rl2 = rb2 + n2, *xlist2 = rl2 + n2, *rc2l = xlist2 + n2;
			int *rh2 = rc2l + n2, *x22 = rh2 + n2, *rf2 = x22 + n2;
			unsigned char *cls2 = (unsigned char*)(rf2 + n2);
			// its own cursors (u64 slots 96.. of the lane's counter block): u32 [0] tier-1 task cursor [1] tier-2 count [2] tier-2 cursor [3] tier-3
			// count [4] tier-3 cursor [7] seed task cursor [10] count of what the larger LDS tier hands on [11] its cursor; slot 102: exported
			// count | k_c2r cursor; slot 103: where its ranks start in the position pool
			unsigned int *q32 = (unsigned int*)(ctr + 96);
			HIPCHK(hipMemcpyAsync(t2, L.rs.sub.data(), n2 * sizeof(bsx_seed_task_t), hipMemcpyHostToDevice, L.st2));
			HIPCHK(hipMemsetAsync(ctr + 96, 0, 80, L.st2));
			HIPCHK(hipMemsetAsync(ctr + 80, 0, 80, L.st2));
			launch_seed(L.st2, g2, d->ix, d_reads, t2, (int)n2, P, (DevIntv*)L.scratch2.p, list_cap, (int)cap2, (DevIntv*)L.out.p, dense_cap, ctr + 4,
			            off2, cnt2, q32 + 7, ctr + SEED3_CTR, 0, (unsigned int*)L.slabflags.p, g2 * 4, 0, 0, (uint32_t*)L.qpack.p);   // (the main launch is over: its packed reads are no longer needed)
			HIPCHK(hipStreamWaitEvent(L.st2, L.rs.ev_tiers, 0));   // the slabs of the HBM tiers and the export pool's lists are shared with the main launch sequence
			RgXPoolArg XB = XA;
			XB.xoff = xoff2; XB.xlist = xlist2; XB.xcount = (unsigned int*)(ctr + 102);
			launch_occ(L.st2, d->n_cu, d->ix, t2, (int)n2, (const DevIntv*)L.out.p, off2, cnt2, opt->max_occ, d_pos, pos_cap, ctr + 11, posoff2, ctr, cls2, ctr + 103);
			if ((rc = run_tiers(L.st2, t2, (int64_t)n2, off2, cnt2, roff2, rn2, ra2, rm2, rb2, q32, (unsigned int*)(ctr + 102), XB, posoff2, cls2, false, rl2, q32 + 6, q32 + 8, (unsigned int*)(ctr + 18), rc2l, (unsigned int*)(ctr + 104), (unsigned int*)(ctr + 105),
				                    rh2, x22, rf2, (unsigned int*)(ctr + 80))) != BSX_OK) return rc;
			HIPCHK(hipMemcpyAsync(L.rs.hres.p, roff2, n2 * 8, hipMemcpyDeviceToHost, L.st2));
			HIPCHK(hipMemcpyAsync((char*)L.rs.hres.p + n2 * 8, rn2, n2 * 4, hipMemcpyDeviceToHost, L.st2));
			HIPCHK(hipEventRecord(L.rs.ev, L.st2));
			L.rs.active = true;
		} else redo.clear();
	}
	clock_gettime(CLOCK_MONOTONIC, &ts2);
	HIPCHK(hipEventRecord(L.ev2, L.st));
	{
		float ms0 = 0, ms1 = 0, ms2 = 0;
		HIPCHK(hipEventSynchronize(L.ev2));
		clock_gettime(CLOCK_MONOTONIC, &ts3);
		if (trace_tiers && n_marks > 1) {
			fprintf(stderr, "[M::regions_batch] region launches (ms):");
			for (int k = 1; k < n_marks; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, L.tier_ev[k - 1], L.tier_ev[k]) == hipSuccess) fprintf(stderr, " %s %.1f |", mark_name[k], ms); }
			if (early3) { float ms = 0; unsigned int ne = 0; D2H(L.st, &ne, ctr + 22, 4); if (hipEventElapsedTime(&ms, L.ev_t3a, L.ev_t3b) == hipSuccess) fprintf(stderr, " tier 3 beside them (%u strand searches) %.1f |", ne, ms); }
			fprintf(stderr, "\n");
		}
		if (trace) fprintf(stderr, "[M::regions_batch] seed kernel done +%.0f ms | redo of %zu strand searches enqueued +%.0f ms | all region tiers done +%.0f ms\n",
		                   (ts1.tv_sec - ts0.tv_sec) * 1e3 + (ts1.tv_nsec - ts0.tv_nsec) * 1e-6, redo.size(), (ts2.tv_sec - ts0.tv_sec) * 1e3 + (ts2.tv_nsec - ts0.tv_nsec) * 1e-6,
		                   (ts3.tv_sec - ts0.tv_sec) * 1e3 + (ts3.tv_nsec - ts0.tv_nsec) * 1e-6);
		HIPCHK(hipEventElapsedTime(&ms0, L.ev0, L.ev1));
		float ms3 = 0, ms_again = 0;
		HIPCHK(hipEventElapsedTime(&ms_again, L.ev5, L.ev6));   // the second seeding pass, when it ran inside this sequence
		if (merged && trace) { // requests per strand search of the second pass, by power of two (k_seedt, prof & 2)
			unsigned long long hh[20];
			HIPCHK(hipMemcpy(hh, ctr + SEED2_CTR + 60, sizeof(hh), hipMemcpyDeviceToHost)); HIPCHK(hipMemset(ctr + SEED2_CTR + 60, 0, sizeof(hh)));
			fprintf(stderr, "[M::regions_batch] second seeding pass %.1f ms (budget %d requests); strand searches by requests made, < 2^k:", ms_again, trip_budget * budget2_mul);
			for (int k = 0; k < 20; ++k) if (hh[k]) fprintf(stderr, " k=%d: %llu |", k, hh[k]);
			fprintf(stderr, "\n");
		}
		if (merged) { L.k_ms[7] += ms_again; L.k_launch[7] += 1; L.seed2_ms += ms_again; L.seed2_launches += 1; L.seed2_tasks += (uint64_t)merged_which.size(); }   // (slot 7: seeding outside the chunk-wide launch)
		HIPCHK(hipEventElapsedTime(&ms3, L.ev6, L.ev4));   // K3 for the chunk (k_occ_expand + k_occ)
		L.k_ms[1] += ms3; L.k_launch[1] += 1;
		HIPCHK(hipEventElapsedTime(&ms1, L.ev4, L.ev3));   // the first region tier alone
		HIPCHK(hipEventElapsedTime(&ms2, L.ev3, L.ev2));   // tiers 2 and 3 and the wait for re-seeded strand searches
		L.k_ms[0] += ms0; L.k_launch[0] += 1; L.k_ms[5] += ms1; L.k_launch[5] += 1; L.k_ms[6] += ms2; L.k_launch[6] += 1;
		HIPCHK(hipGetLastError());
	}
	unsigned long long used = 0;
	std::vector<long long> h_off((size_t)n);
	D2H(L.st, h_off.data(), r_off, (size_t)n * 8);
	D2H(L.st, out_n, r_n, (size_t)n * 4);
	D2H(L.st, &used, ctr + 6, 8);
	if (used > regs_cap) used = regs_cap;
	{ // work of the chunk (bsx_device_region_work)
		unsigned long long n_occ = 0;
		D2H(L.st, &n_occ, ctr + 11, 8);
		L.work[0] += (uint64_t)n; L.work[2] += n_occ < pos_cap ? n_occ : pos_cap; L.work[3] += used;
		for (int64_t i = 0; i < n; ++i) L.work[4] += (uint64_t)(tasks[i].len > 0 ? tasks[i].len : 0);
	}
	for (int64_t i = 0; i < n; ++i) out_off[i] = h_off[i];
	L.rs.used_main = used; L.rs.regs_cap = regs_cap;
	for (size_t j = 0; j < redo.size(); ++j) out_n[redo[j]] = BSX_REGIONS_PENDING;   // lane_regions_finish fills these in
	if (*out_cap < (int64_t)used + 65536) { *out_cap = (int64_t)used + 65536; *out = (bsx_region_t*)realloc(*out, sizeof(bsx_region_t) * (size_t)*out_cap); }
	D2H(L.st, *out, L.regs.p, (size_t)used * sizeof(bsx_region_t));

	if (trace) {
		unsigned int hc[12]; unsigned long long hu[12];
		D2H(L.st, hc, c32, sizeof(hc));
		D2H(L.st, hu, ctr, sizeof(hu));
		unsigned long long hu20 = 0;
		D2H(L.st, &hu20, ctr + 20, 8);
		fprintf(stderr, "[M::regions_batch] %lld strand searches: %llu intervals, %llu occurrences looked up ahead | left tier 1: %u, left tier 1b: %u, reached tier 2: %u, left tier 2: %u | second chains -> regions launch: %u | every tier exports: %d\n",
		        (long long)n, hu[4], hu[11], hc[1], hc[10], long_reads || export_all ? hc[6] : hc[6] ? hc[6] : hc[10], hc[3], (unsigned int)hu20, export_all ? 1 : 0);
		unsigned long long sp[8];
		D2H(L.st, sp, ctr + 48, sizeof(sp));
		HIPCHK(hipMemsetAsync(ctr + 48, 0, sizeof(sp), L.st));
		if (sp[0]) fprintf(stderr, "[M::regions_batch] k_seed: %.0f M wave cycles, %.1f%% in the full machine (%llu passes, %.0f cycles each), publishing %.1f%% | %llu wave trips, %.0f cycles per trip\n",
		                   sp[0] * 1e-6, 100.0 * sp[1] / sp[0], sp[4], sp[4] ? (double)sp[1] / sp[4] : 0.0, 100.0 * sp[2] / sp[0], sp[3], sp[3] ? (double)sp[0] / sp[3] : 0.0);
		{
			unsigned long long tq[4];
			D2H(L.st, tq, ctr + 121, sizeof(tq));
			HIPCHK(hipMemsetAsync(ctr + 121, 0, sizeof(tq), L.st));
			if (sp[0] && tq[3]) fprintf(stderr, "[M::regions_batch] k_seedt: short machine %.1f%%, fetch %.1f%%, post %.1f%% of the wave cycles | %.1f requests per wave trip\n",
			                            100.0 * tq[0] / sp[0], 100.0 * tq[1] / sp[0], 100.0 * tq[2] / sp[0], sp[3] ? (double)tq[3] / sp[3] : 0.0);
		}
		{
			unsigned long long xp[11];
			D2H(L.st, xp, ctr + 56, sizeof(xp));
			HIPCHK(hipMemsetAsync(ctr + 56, 0, sizeof(xp), L.st));
			if (xp[0] || xp[6]) fprintf(stderr, "[M::regions_batch] k_ext4: %llu jobs, %llu rows in %llu wave trips (%.2f rows per trip), %llu passes between extensions, %.2f slots per trip, %llu rows of narrow jobs\n", xp[0], xp[1], xp[2], xp[2] ? (double)xp[1] / xp[2] : 0.0, xp[3], xp[2] ? (double)xp[4] / xp[2] : 0.0, xp[5]);
			if (xp[6]) fprintf(stderr, "[M::regions_batch] k_extl: %llu jobs (%llu sent on to k_ext4), %llu rows in %llu wave trips (%.1f rows per trip), %llu passes between extensions\n", xp[6], xp[10], xp[7], xp[8], xp[8] ? (double)xp[7] / xp[8] : 0.0, xp[9]);
		}
		unsigned long long pf[16];
		D2H(L.st, pf, ctr + 32, sizeof(pf));
		HIPCHK(hipMemsetAsync(ctr + 32, 0, sizeof(pf), L.st));
		double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)pf[k];
		if (tot > 0) fprintf(stderr, "[M::regions_batch] wave cycles by stage (all tiers): intervals %.1f%% occurrences %.1f%% chaining %.1f%% weights+order %.1f%% sort %.1f%% filter %.1f%% chain prologues+seed tests %.1f%% extension %.1f%% | %.0f M cycles, %llu extensions, %llu rows\n",
		        100 * pf[0] / tot, 100 * pf[1] / tot, 100 * pf[2] / tot, 100 * pf[3] / tot, 100 * pf[4] / tot, 100 * pf[5] / tot, 100 * pf[6] / tot, 100 * pf[7] / tot, tot * 1e-6, pf[8], pf[9]);
		if (pf[10]) fprintf(stderr, "[M::regions_batch] seed filter: %llu alignments\n", pf[10]);
		if (pf[11]) fprintf(stderr, "[M::regions_batch] seed loops (mem_chain2region1): %llu seeds reached, %llu skipped as contained, %llu took the extension made ahead, %llu extended in place\n", pf[11], pf[12], pf[13], pf[14]);
		{ // the HBM tiers: how long their strand searches take (a wave each)
			unsigned long long tk[8];
			D2H(L.st, tk, ctr + 110, sizeof(tk));
			HIPCHK(hipMemsetAsync(ctr + 110, 0, sizeof(tk), L.st));
			for (int k = 0; k < 2; ++k) if (tk[4 * k + 2])
				fprintf(stderr, "[M::regions_batch] tier %d: %llu strand searches, %.2f M cycles each on average, the longest %.1f M, the busiest wave %.1f M, all of them %.0f M\n",
				        2 + k, tk[4 * k + 2], 1e-6 * tk[4 * k + 1] / tk[4 * k + 2], 1e-6 * tk[4 * k], 1e-6 * tk[4 * k + 3], 1e-6 * tk[4 * k + 1]);
			unsigned long long th[24];
			D2H(L.st, th, ctr + 128, sizeof(th));
			HIPCHK(hipMemsetAsync(ctr + 128, 0, sizeof(th), L.st));
			if (th[0]) {
				fprintf(stderr, "[M::regions_batch] tier 3: the longest strand search: %.1f M cycles, %llu intervals, %llu chains kept, %llu regions | strand searches by duration, < 2^k x 65536 cycles:",
				        (double)(th[0] >> 36) * 4096e-6, (th[0] >> 24) & 4095, (th[0] >> 14) & 16383, th[0] & 16383);
				for (int k = 0; k < 22; ++k) if (th[2 + k]) fprintf(stderr, " k=%d: %llu |", k, th[2 + k]);
				fprintf(stderr, "\n");
			}
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &ts_out);
	if (trace) fprintf(stderr, "[M::regions_batch] entry to kernels enqueued %.0f ms | tiers done to regions downloaded %.0f ms (%llu regions)\n",
	                   (ts0.tv_sec - ts_in.tv_sec) * 1e3 + (ts0.tv_nsec - ts_in.tv_nsec) * 1e-6, (ts_out.tv_sec - ts3.tv_sec) * 1e3 + (ts_out.tv_nsec - ts3.tv_nsec) * 1e-6, used);
	// declined tasks: hand their interval lists back (ordered by info, as bsx_seed_batch returns them)
	std::vector<int64_t> decl;
	for (int64_t i = 0; i < n; ++i) if (out_n[i] < -1 && out_n[i] != BSX_REGIONS_PENDING) decl.push_back(i);
	decl_off[0] = 0;
	if (!decl.empty()) {
		std::vector<long long> s_off((size_t)n); std::vector<int> s_n((size_t)n);
		D2H(L.st, s_off.data(), d_off, (size_t)n * 8);
		D2H(L.st, s_n.data(), d_n, (size_t)n * 4);
		int64_t tot = 0;
		for (size_t j = 0; j < decl.size(); ++j) { decl_off[j] = tot; tot += s_n[decl[j]]; }
		decl_off[decl.size()] = tot;
		if (*decl_cap < tot) { *decl_cap = tot + (tot >> 2) + 16; *decl_intv = (bsx_intv_t*)realloc(*decl_intv, sizeof(bsx_intv_t) * (size_t)*decl_cap); }
		const bsx_intv_t *dense = nullptr;
		if (decl.size() > 256 && tot > 0) { // many: their lists gathered on the device (they lie a strand search's stretch apart), one copy
			std::vector<long long> which(decl.begin(), decl.end()), doff(decl_off, decl_off + decl.size());
			for (size_t j = 0; j < decl.size(); ++j) if (s_n[decl[j]] < 0) { which.clear(); break; }   // (a list that did not fit has no entries to fetch: leave the one-by-one path to skip it)
			if (!which.empty()) {
				const size_t nb = decl.size() * 8;
				if ((rc = L.gath.reserve(2 * nb + (size_t)tot * sizeof(bsx_intv_t) + 64)) != BSX_OK) return rc;
				H2D(L.st, L.gath.p, which.data(), nb);
				H2D(L.st, (char*)L.gath.p + nb, doff.data(), nb);
				DevIntv *gd = (DevIntv*)((char*)L.gath.p + 2 * nb);
				launch_gather_lists(L.st, (const DevIntv*)L.out.p, d_off, d_n, (const long long*)L.gath.p, (const long long*)((char*)L.gath.p + nb), (long long)decl.size(), gd);
				if ((rc = L.hstage.reserve((size_t)tot * sizeof(bsx_intv_t) + 64)) != BSX_OK) return rc;
				D2H(L.st, L.hstage.p, gd, (size_t)tot * sizeof(bsx_intv_t));
				dense = (const bsx_intv_t*)L.hstage.p;
			}
		}
		for (size_t j = 0; j < decl.size(); ++j) {
			const int64_t i = decl[j]; const int cnt = s_n[i];
			bsx_intv_t *dst = *decl_intv + decl_off[j];
			if (cnt <= 0) continue;
			if (dense) memcpy(dst, dense + decl_off[j], sizeof(bsx_intv_t) * (size_t)cnt);
			else D2H(L.st, dst, (const bsx_intv_t*)L.out.p + s_off[i], sizeof(bsx_intv_t) * (size_t)cnt);
			if (cnt > 1) std::sort(dst, dst + cnt, intv_info_lt);
		}
	}
	if (trace) { // one line per call, with the lane: lines of chunks in flight together interleave
		struct timespec te; clock_gettime(CLOCK_MONOTONIC, &te);
#define MS_(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
		fprintf(stderr, "[M::regions_batch] lane %d from %.3f: %.0f ms = enqueue %.0f + seeding awaited %.0f + second pass enqueued %.0f + tiers awaited %.0f + counts and regions down %.0f + the rest %.0f\n",
		        lane, ts_in.tv_sec % 1000 + ts_in.tv_nsec * 1e-9, MS_(ts_in, te), MS_(ts_in, ts0), MS_(ts0, ts1), MS_(ts1, ts2), MS_(ts2, ts3), MS_(ts3, ts_out), MS_(ts_out, te));
#undef MS_
	}
	return BSX_OK;
}

// the strand searches lane_regions_batch left pending: their regions are appended to *out, out_off/out_n filled in
// (out_n = -1: even the longest lists or the largest tier did not hold them, the caller seeds and chains them itself)
static int lane_regions_finish(bsx_device_t *d, int lane, bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (!L.rs.active) return BSX_OK;
	HIPCHK(hipSetDevice(d->ordinal));
	struct timespec ta, tb;
	clock_gettime(CLOCK_MONOTONIC, &ta);
	HIPCHK(hipEventSynchronize(L.rs.ev));
	clock_gettime(CLOCK_MONOTONIC, &tb);
	if (bsx_phases()) fprintf(stderr, "[M::regions_finish] waited %.0f ms for %zu strand searches seeded again\n", (tb.tv_sec - ta.tv_sec) * 1e3 + (tb.tv_nsec - ta.tv_nsec) * 1e-6, L.rs.tasks.size());
	HIPCHK(hipGetLastError());
	const size_t n2 = L.rs.tasks.size();
	const long long *roff = (const long long*)L.rs.hres.p;
	const int *rn = (const int*)((const char*)L.rs.hres.p + n2 * 8);
	int64_t used = (int64_t)L.rs.used_main;
	// their regions lie behind the main sequence's in the device's pool: one copy of that stretch (a copy per strand search was tens
	// of thousands of small transfers per chunk on a repeat-rich genome), then each list is taken from it
	unsigned long long used_all = 0;
	D2H(L.st2, &used_all, dev_counters(L) + 6, 8);
	if (used_all > L.rs.regs_cap) used_all = L.rs.regs_cap;   // (a cursor past the pool: the strand searches that found no room carry status 7)
	const unsigned long long lo = L.rs.used_main, hi = std::max<unsigned long long>(used_all, lo);
	std::vector<bsx_region_t> stretch;
	if (hi > lo) {
		bool any = false;
		for (size_t j = 0; j < n2 && !any; ++j) any = rn[j] > 0;
		if (any) { stretch.resize((size_t)(hi - lo)); D2H(L.st2, stretch.data(), (const bsx_region_t*)L.regs.p + lo, sizeof(bsx_region_t) * (size_t)(hi - lo)); }
	}
	long hist[16] = {0};
	for (size_t j = 0; j < n2; ++j) {
		const int64_t i = L.rs.tasks[j];
		if (rn[j] < 0) { out_n[i] = -1; out_off[i] = 0; ++hist[(-rn[j]) & 15]; continue; }
		if (*out_cap < used + rn[j]) { *out_cap = used + rn[j] + 1024 + (*out_cap >> 2); *out = (bsx_region_t*)realloc(*out, sizeof(bsx_region_t) * (size_t)*out_cap); }
		if (rn[j] > 0) {
			if (!stretch.empty() && (unsigned long long)roff[j] >= lo && (unsigned long long)roff[j] + (unsigned long long)rn[j] <= hi)
				memcpy(*out + used, stretch.data() + ((unsigned long long)roff[j] - lo), sizeof(bsx_region_t) * (size_t)rn[j]);
			else D2H(L.st2, *out + used, (const bsx_region_t*)L.regs.p + roff[j], sizeof(bsx_region_t) * (size_t)rn[j]);
		}
		out_off[i] = used; out_n[i] = rn[j];
		used += rn[j];
	}
	if (bsx_phases()) {
		fprintf(stderr, "[M::regions_finish] left to the caller after the second pass, by reason:");
		for (int k = 1; k < 16; ++k) if (hist[k]) fprintf(stderr, " %d: %ld", k, hist[k]);
		fprintf(stderr, "\n");
	}
	L.rs.active = false;
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------
static int lane_sa_batch(bsx_device_t *d, int lane, int64_t n, const bsx_sa_job_t *jobs, uint64_t *pos)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) return BSX_OK;
	HIPCHK(hipSetDevice(d->ordinal));
	int rc;
	if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_sa_job_t))) != BSX_OK) return rc;
	if ((rc = L.res.reserve((size_t)n * 8)) != BSX_OK) return rc;
	HIPCHK(hipMemcpyAsync(L.jobs.p, jobs, (size_t)n * sizeof(bsx_sa_job_t), hipMemcpyHostToDevice, L.st));
	int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)d->n_cu * 8);
	HIPCHK(hipEventRecord(L.ev0, L.st));
	launch_sa(L.st, grid, d->ix, (const bsx_sa_job_t*)L.jobs.p, (long long)n, (uint64_t*)L.res.p, dev_counters(L));
	HIPCHK(hipEventRecord(L.ev1, L.st));
	if ((rc = finish_timed(L, 1)) != BSX_OK) return rc;
	D2H(L.st, pos, L.res.p, (size_t)n * 8);
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K4
// ------------------------------------------------------------------------------------------
static int lane_extend_batch(bsx_device_t *d, int lane, int64_t n, const bsx_ext_job_t *jobs, bsx_ext_res_t *res)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) return BSX_OK;
	HIPCHK(hipSetDevice(d->ordinal));
	if (bsx_tune_is_set("ext4")) { // tests: the batch through the quarter-wave kernel of the regions path (k_ext4.hip); jobs it declines fail the call
		int rc, max_q = 0;
		for (int64_t i = 0; i < n; ++i) max_q = std::max(max_q, jobs[i].qlen);
		if ((max_q > x4_max_query(16) && bsx_tune_long("ext4", 0) != 3) || n > 0x7fffffff) return BSX_E_ARG;
		if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_ext_job_t))) != BSX_OK) return rc;
		if ((rc = L.res.reserve((size_t)n * sizeof(bsx_ext_res_t))) != BSX_OK) return rc;
		if ((rc = L.aux.reserve(64)) != BSX_OK) return rc;
		HIPCHK(hipMemcpyAsync(L.jobs.p, jobs, (size_t)n * sizeof(bsx_ext_job_t), hipMemcpyHostToDevice, L.st));
		HIPCHK(hipMemsetAsync(L.aux.p, 0, 64, L.st));
		if (bsx_tune_long("ext4", 0) != 3) launch_ext4_batch(L.st, d->n_cu, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_ext_job_t*)L.jobs.p, (bsx_ext_res_t*)L.res.p, (unsigned int)n, (unsigned int*)L.aux.p, max_q);
		if (bsx_tune_long("ext4", 0) == 3) { // ... or the wavefront-per-job form whose rows follow the band in a register window (ext_dp_win: the long reads' chains -> regions launch)
			launch_extwin_batch(L.st, d->n_cu, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_ext_job_t*)L.jobs.p, (bsx_ext_res_t*)L.res.p, (long long)n);
		} else
		if (bsx_tune_long("ext4", 0) == 2) { // then the lane-per-job kernel (k_extl.hip) over the same jobs: its answers replace the others'
			HIPCHK(hipMemsetAsync(L.aux.p, 0, 64, L.st));
			launch_extl_batch(L.st, d->n_cu, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_ext_job_t*)L.jobs.p, (bsx_ext_res_t*)L.res.p, (unsigned int)n, (unsigned int*)L.aux.p);
			unsigned int c[4]; D2H(L.st, c, L.aux.p, 16);
			fprintf(stderr, "[M::extl] %lld jobs, %u left to k_ext4\n", (long long)n, c[0]);
		}
		HIPCHK(hipGetLastError());
		D2H(L.st, res, L.res.p, (size_t)n * sizeof(bsx_ext_res_t));
		for (int64_t i = 0; i < n; ++i) if (res[i].score == X4_DECLINED) return BSX_E_ARG;
		return BSX_OK;
	}
	// classes by LDS footprint (query length) and row width (band): {qcap, max band columns, NC}
	// The host chaining path calls this once per round of its strand searches' extensions -- a thousand rounds of a few hundred jobs for
	// the reads inside satellite arrays of a repeat-rich genome -- while the front-half kernels of other chunks fill the device.
	// $BSX_HOSTPATH_STREAM=1 puts the rounds on the lane's high-priority stream: measured, no difference (3.36-3.60 against 3.52 s per chunk).
	const int hp_hi = 0;
	hipStream_t S = hp_hi ? L.st_hi : L.st;
	// the last class: queries of any length, rows in HBM (reads of tens of kilobases).  What remains out of reach is a band of more
	// than 2048 columns, i.e. -w above 511 -- a limit of the option, not of the data
	static const int QCAP[4] = {256, 1024, 16384, 0x7fffffff}, BAND[4] = {256, 512, 2048, 2048}, NCS[4] = {4, 8, 32, 32};
	std::vector<int> order[4];
	int hbm_qmax = 0, rc0;
	for (int64_t i = 0; i < n; ++i) {
		const bsx_ext_job_t &j = jobs[i];
		if (j.qlen <= 0 || j.tlen < 0 || j.h0 <= 0) { fprintf(stderr, "[bsx-hip] extend job %lld: invalid (qlen=%d tlen=%d h0=%d)\n", (long long)i, j.qlen, j.tlen, j.h0); return BSX_E_ARG; }
		long long band = std::min<long long>(j.qlen, 2LL * j.w + 1);
		int c = 0;
		while (c < 4 && (j.qlen > QCAP[c] || band > BAND[c])) ++c;
		if (c == 4) { fprintf(stderr, "[bsx-hip] extend job %lld: a band of %lld columns is beyond the kernel's 2048 (-w above 511)\n", (long long)i, band); return BSX_E_ARG; }
		if (c == 3) hbm_qmax = std::max(hbm_qmax, j.qlen);
		order[c].push_back((int)i);
	}
	const int hbm_blocks = order[3].empty() ? 0 : (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((order[3].size() + 3) / 4, (size_t)d->n_cu), ((size_t)2 << 30) / (4 * extend_hbm_row_bytes(hbm_qmax))));
	if (hbm_blocks && (rc0 = L.scratch.reserve((size_t)hbm_blocks * 4 * extend_hbm_row_bytes(hbm_qmax))) != BSX_OK) return rc0;
	int rc;
	if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_ext_job_t))) != BSX_OK) return rc;
	if ((rc = L.res.reserve((size_t)n * sizeof(bsx_ext_res_t))) != BSX_OK) return rc;
	if ((rc = L.aux.reserve((size_t)n * 4 + 64)) != BSX_OK) return rc;
	HIPCHK(hipMemcpyAsync(L.jobs.p, jobs, (size_t)n * sizeof(bsx_ext_job_t), hipMemcpyHostToDevice, S));
	size_t off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		HIPCHK(hipMemcpyAsync((int*)L.aux.p + off, order[c].data(), order[c].size() * 4, hipMemcpyHostToDevice, S));
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev0, S));
	off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		if (c == 3)
			launch_extend_hbm(S, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_ext_job_t*)L.jobs.p, (const int*)L.aux.p + off,
			                  (long long)order[c].size(), (bsx_ext_res_t*)L.res.p, hbm_qmax, hbm_blocks, L.scratch.p);
		else
		launch_extend(S, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_ext_job_t*)L.jobs.p, (const int*)L.aux.p + off,
		              (long long)order[c].size(), (bsx_ext_res_t*)L.res.p, QCAP[c], NCS[c], d->n_cu);
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev1, S));
	if ((rc = finish_timed(L, 2)) != BSX_OK) return rc;
	D2H(S, res, L.res.p, (size_t)n * sizeof(bsx_ext_res_t));
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K5
// ------------------------------------------------------------------------------------------
static int lane_sw_batch(bsx_device_t *d, int lane, int64_t n, const bsx_sw_job_t *jobs, bsx_sw_res_t *res)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) return BSX_OK;
	std::lock_guard<std::mutex> hi_lock(L.hi_mu);
	HIPCHK(hipSetDevice(d->ordinal));
	// byte-sized jobs (KSW_XBYTE, queries of up to 256 columns: mate rescue of ordinary reads) go four to a wavefront through the striped
	// kernel (k_swl.hip), ordered by stripe count and then by target length, longest first, so that the four jobs of a wavefront are alike;
	// the others a wavefront each (k_sw.hip), by padded query length: up to 256, 1024, 3072 columns
	std::vector<int> order[4];
	int max_tlen = 1, slen_max = 1;
	const bool use_swl = true;
	std::vector<int> key;
	for (int64_t i = 0; i < n; ++i) {
		const bsx_sw_job_t &j = jobs[i];
		if (j.qlen <= 0 || j.tlen < 0) { fprintf(stderr, "[bsx-hip] sw job %lld: invalid\n", (long long)i); return BSX_E_ARG; }
		const int p = (j.xtra & BSX_KSW_XBYTE) ? 16 : 8, Q = (j.qlen + p - 1) / p * p;
		if (Q > 3072) { fprintf(stderr, "[bsx-hip] sw job %lld: query %d beyond kernel limit (3072)\n", (long long)i, j.qlen); return BSX_E_ARG; }
		max_tlen = std::max(max_tlen, j.tlen);
		if (use_swl && p == 16 && Q <= 256) { order[3].push_back((int)i); slen_max = std::max(slen_max, Q / 16); }
		else order[Q <= 256 ? 0 : Q <= 1024 ? 1 : 2].push_back((int)i);
	}
	if (order[3].size() > 4) { // counting sort by (stripes, target length descending)
		const int NK = 17 * 2048;
		std::vector<int> cnt((size_t)NK + 1, 0), sorted(order[3].size());
		auto keyof = [&](int i) { const bsx_sw_job_t &j = jobs[i]; return ((j.qlen + 15) >> 4) * 2048 + (2047 - std::min(j.tlen, 2047)); };
		for (int i : order[3]) ++cnt[(size_t)keyof(i) + 1];
		for (int x = 0; x < NK; ++x) cnt[(size_t)x + 1] += cnt[(size_t)x];
		for (int i : order[3]) sorted[(size_t)cnt[(size_t)keyof(i)]++] = i;
		order[3].swap(sorted);
	}
	int rc;
	const int blocks_cap = d->n_cu * 8;
	if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_sw_job_t))) != BSX_OK) return rc;
	if ((rc = L.res.reserve((size_t)n * sizeof(bsx_sw_res_t))) != BSX_OK) return rc;
	if ((rc = L.aux.reserve((size_t)n * 4 + 64)) != BSX_OK) return rc;
	if ((rc = L.scratch.reserve((size_t)blocks_cap * 4 * 4 * (size_t)max_tlen * 8)) != BSX_OK) return rc;   // b[] of every job in flight: four jobs to a wave in k_swl
	const bool tr_sw = bsx_phases() != 0;
	const double ts0 = tr_sw ? bsx_now_s() : 0;
	H2D(L.st_hi, L.jobs.p, jobs, (size_t)n * sizeof(bsx_sw_job_t));
	size_t off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		H2D(L.st_hi, (int*)L.aux.p + off, order[c].data(), order[c].size() * 4);
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev0, L.st_hi));
	off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		const long long m = (long long)order[c].size();
		if (c == 3) {
			const int blocks = (int)std::min<long long>((m + 15) / 16, blocks_cap);
			launch_swl(L.st_hi, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_sw_job_t*)L.jobs.p, (const int*)L.aux.p + off, m,
			           (bsx_sw_res_t*)L.res.p, (unsigned long long*)L.scratch.p, max_tlen, blocks, slen_max);
		} else {
			const int blocks = (int)std::min<long long>((m + 3) / 4, blocks_cap);
			launch_sw(L.st_hi, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_sw_job_t*)L.jobs.p, (const int*)L.aux.p + off, m,
			          (bsx_sw_res_t*)L.res.p, (unsigned long long*)L.scratch.p, max_tlen, blocks, c == 0 ? 4 : c == 1 ? 16 : 48);
		}
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev1, L.st_hi));
	const double ts1 = tr_sw ? bsx_now_s() : 0;
	if ((rc = finish_timed(L, 3)) != BSX_OK) return rc;
	const double ts2 = tr_sw ? bsx_now_s() : 0;
	D2H(L.st_hi, res, L.res.p, (size_t)n * sizeof(bsx_sw_res_t));
	if (tr_sw) { float ms = 0; (void)hipEventElapsedTime(&ms, L.ev0, L.ev1);
		fprintf(stderr, "[M::sw_batch] %lld jobs: upload %.3f s, waited %.3f s for the kernels (%.3f s on the device), download %.3f s\n", (long long)n, ts1 - ts0, ts2 - ts1, ms * 1e-3, bsx_now_s() - ts2); }
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// C5 over the regions of the last regions batch of this lane
// ------------------------------------------------------------------------------------------
static int lane_regions_dedup(bsx_device_t *d, int lane, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx,
                              int64_t *long_off = nullptr, uint16_t **long_idx = nullptr, int64_t *long_cap = nullptr)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	if (!opt || !out_n || !out_idx || per_read < 1) return BSX_E_ARG;
	Lane &L = d->lane[lane];
	if (n_reads == 0) return BSX_OK;
	if (n_reads * per_read != L.rb_tasks) { fprintf(stderr, "[bsx-hip] regions_dedup: %lld reads x %d strand searches, but the last regions batch had %lld\n", (long long)n_reads, per_read, (long long)L.rb_tasks); return BSX_E_ARG; }
	HIPCHK(hipSetDevice(d->ordinal));
	int rc;
	const int64_t n = L.rb_tasks;
	const size_t cap = (size_t)dedup_cap();
	// the reads with more regions than a lane of k_dedup holds (long_off given): a wavefront each (k_dedup_long), their lists of 16-bit indices
	// one behind the other in a pool with room for every region the main sequence made
	const bool with_long = long_off && long_idx && long_cap && per_read <= 4 && bsx_tune_long("long_dedup", 1) != 0;
	const size_t pool_cap = with_long ? (size_t)L.rs.used_main + 64 : 0;
	// layout of L.dd: counts | short lists | [long: class lists (2 n_reads ints) | offsets (n_reads i64) | 2 counters + cursor (16 B) | pool]
	const size_t o_idx = (size_t)n_reads * 4, o_list = (o_idx + (size_t)n_reads * cap + 15) & ~(size_t)15, o_off = o_list + (size_t)n_reads * 8, o_ctr = o_off + (size_t)n_reads * 8, o_pool = o_ctr + 16;
	if ((rc = L.dd.reserve(with_long ? o_pool + pool_cap * 2 + 64 : o_idx + (size_t)n_reads * cap + 64)) != BSX_OK) return rc;
	const long long *r_off = (const long long*)L.regmeta.p; const int *r_n = (const int*)((const char*)L.regmeta.p + (size_t)n * 8);
	int *d_n = (int*)L.dd.p; unsigned char *d_idx = (unsigned char*)L.dd.p + o_idx;
	int *d_list = with_long ? (int*)((char*)L.dd.p + o_list) : nullptr;
	long long *d_off = (long long*)((char*)L.dd.p + o_off);
	unsigned int *d_cnt = with_long ? (unsigned int*)((char*)L.dd.p + o_ctr) : nullptr;
	if (with_long) {
		HIPCHK(hipMemsetAsync(d_cnt, 0, 16, L.st));
		HIPCHK(hipMemsetAsync(d_off, 0xff, (size_t)n_reads * 8, L.st));   // -1: the read's list (if it has one) is among the short ones
	}
	L.ddl.valid = true; L.ddl.n_reads = n_reads; L.ddl.per_read = per_read; L.ddl.o_idx = o_idx; L.ddl.o_off = o_off; L.ddl.o_pool = o_pool; L.ddl.with_long = with_long;
	launch_dedup(L.st, (const bsx_region_t*)L.regs.p, r_off, r_n, (int)n_reads, per_read, (long long)d->ix.l_pac, opt->max_chain_gap, opt->w, opt->mask_level_redun, d_n, d_idx, d_list, d_cnt);
	if (with_long)
		launch_dedup_long(L.st, d->n_cu, (const bsx_region_t*)L.regs.p, r_off, r_n, (int)n_reads, per_read, (long long)d->ix.l_pac, opt->max_chain_gap, opt->w, opt->mask_level_redun,
		                  d_list, d_cnt, d_n, d_off, (unsigned short*)((char*)L.dd.p + o_pool), (unsigned long long)pool_cap, (unsigned long long*)(d_cnt + 2));
	HIPCHK(hipGetLastError());
	D2H(L.st, out_n, d_n, (size_t)n_reads * 4);
	D2H(L.st, out_idx, d_idx, (size_t)n_reads * cap);
	if (with_long) {
		unsigned int hc[4];
		D2H(L.st, hc, d_cnt, 16);
		unsigned long long used = (unsigned long long)hc[2] | (unsigned long long)hc[3] << 32;
		if (used > pool_cap) used = pool_cap;   // (lists that found no room left their reads to the caller)
		D2H(L.st, long_off, d_off, (size_t)n_reads * 8);
		if (*long_cap < (int64_t)used + 16) { *long_cap = (int64_t)used + (int64_t)(used >> 2) + 1024; *long_idx = (uint16_t*)realloc(*long_idx, (size_t)*long_cap * 2); }
		if (used) D2H(L.st, *long_idx, (char*)L.dd.p + o_pool, (size_t)used * 2);
		if (bsx_phases()) fprintf(stderr, "[M::regions_dedup] a wavefront per read: %u reads of up to 256 regions, %u of up to %d; %llu regions kept\n", hc[0], hc[1], dedup_long_cap(), used);
	} else if (long_off) for (int64_t i = 0; i < n_reads; ++i) long_off[i] = -1;
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// mate rescue's plan + its K5 batch on the device (k_msw.hip), for the pairs [p0, p1) of the chunk whose reads the lane last de-duplicated.
// Returns BSX_OK and *n_jobs >= 0, or *n_jobs = -1 when the plan does not apply (the caller plans on the host as before): no
// de-duplication state for these reads, reads too long for the byte-sized kernel, more than 64 candidates a read.
// ------------------------------------------------------------------------------------------
static int lane_msw_plan(bsx_device_t *d, int lane, const bsx_opt_t *opt, const bsx_pestat_t *pes, int64_t token, int64_t n_reads, int per_read,
                         const uint32_t *roff, int max_len, int p0, int p1, void *table, bsx_sw_res_t **res, int64_t *res_cap, int64_t *n_jobs)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	*n_jobs = -1;
	if (p1 <= p0) { *n_jobs = 0; return BSX_OK; }
	if (!bsx_tune_long("msw_plan", 0)) return BSX_OK;   // (built, bit-identical, measured slower than the host's plan pass: off unless asked for -- DESIGN.md section 4)
	if (!L.ddl.valid || L.ddl.n_reads != n_reads || L.ddl.per_read != per_read || n_reads * per_read != L.rb_tasks) return BSX_OK;
	if (max_len <= 0 || max_len > 256 || (long long)max_len * opt->a >= 250 || opt->max_matesw > 64 || opt->max_matesw < 1 || (opt->flag & BSX_F_SELF_OVLP)) return BSX_OK;
	std::lock_guard<std::mutex> hi_lock(L.hi_mu);
	HIPCHK(hipSetDevice(d->ordinal));
	int rc;
	const int np = p1 - p0, MM = opt->max_matesw;
	const size_t job_cap = (size_t)np * 2 * (size_t)MM;
	const size_t tb = (size_t)np * msw_pair_bytes(), o_hist = (tb + 255) & ~(size_t)255, o_cnt = o_hist + (size_t)msw_hist_bins() * 4, o_ord = o_cnt + 256;
	if ((rc = L.msw_jobs.reserve(job_cap * sizeof(bsx_sw_job_t) + 64)) != BSX_OK) return rc;
	if ((rc = L.msw_res.reserve(job_cap * sizeof(bsx_sw_res_t) + 64)) != BSX_OK) return rc;
	if ((rc = L.msw_meta.reserve(o_ord + job_cap * 4 + 64)) != BSX_OK) return rc;
	if (L.msw_token != token) { // the chunk's read offsets, once per chunk
		if ((rc = L.msw_roff.reserve(((size_t)n_reads + 1) * 4)) != BSX_OK) return rc;
		H2D(L.st_hi, L.msw_roff.p, roff, ((size_t)n_reads + 1) * 4);
		L.msw_token = token;
	}
	char *M = (char*)L.msw_meta.p;
	unsigned int *d_hist = (unsigned int*)(M + o_hist), *d_cnt = (unsigned int*)(M + o_cnt);
	int *d_ord = (int*)(M + o_ord);
	HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)msw_hist_bins() * 4 + 256, L.st_hi));
	const int64_t n = L.rb_tasks;
	const long long *r_off = (const long long*)L.regmeta.p; const int *r_n = (const int*)((const char*)L.regmeta.p + (size_t)n * 8);
	const char *DD = (const char*)L.dd.p;
	launch_msw_plan(L.st_hi, d->n_cu, d->ix, (const bsx_region_t*)L.regs.p, r_off, r_n, (const int*)DD, (const unsigned char*)DD + L.ddl.o_idx,
	                L.ddl.with_long ? (const long long*)(DD + L.ddl.o_off) : nullptr, L.ddl.with_long ? (const unsigned short*)(DD + L.ddl.o_pool) : nullptr, dedup_cap(), per_read,
	                (const unsigned int*)L.msw_roff.p, pes->low, pes->high, opt->pen_unpaired, MM, opt->min_seed_len, opt->a, p0, np,
	                (bsx_sw_job_t*)L.msw_jobs.p, (unsigned int)std::min<size_t>(job_cap, 0xfffffff0u), d_cnt, d_hist, M, d_ord);
	unsigned int nj = 0;
	D2H(L.st_hi, &nj, d_cnt, 4);
	if ((size_t)nj > job_cap) nj = (unsigned int)job_cap;
	if (nj) {
		const int blocks_cap = d->n_cu * 8;
		long long bound = (long long)pes->high - (long long)pes->low + max_len + 8;   // no window is longer: re - rb <= high - low + l_ms
		if (bound < 1) bound = 1;
		if (bound > (1 << 20)) { *n_jobs = -1; return BSX_OK; }                       // (absurd insert-size bounds: the host's path sizes its scratch from the jobs)
		if ((rc = L.scratch.reserve((size_t)blocks_cap * 4 * 4 * (size_t)bound * 8)) != BSX_OK) return rc;
		const int blocks = (int)std::min<long long>(((long long)nj + 15) / 16, blocks_cap);
		HIPCHK(hipEventRecord(L.ev0, L.st_hi));
		launch_swl(L.st_hi, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_sw_job_t*)L.msw_jobs.p, (const int*)d_ord, (long long)nj,
		           (bsx_sw_res_t*)L.msw_res.p, (unsigned long long*)L.scratch.p, (int)bound, blocks, (max_len + 15) / 16);
		HIPCHK(hipEventRecord(L.ev1, L.st_hi));
		if ((rc = finish_timed(L, 3)) != BSX_OK) return rc;
		if (*res_cap < (int64_t)nj) { *res_cap = (int64_t)nj + (nj >> 2) + 1024; *res = (bsx_sw_res_t*)realloc(*res, sizeof(bsx_sw_res_t) * (size_t)*res_cap); }
		D2H(L.st_hi, *res, L.msw_res.p, (size_t)nj * sizeof(bsx_sw_res_t));
	}
	D2H(L.st_hi, table, M, tb);
	HIPCHK(hipGetLastError());
	*n_jobs = (int64_t)nj;
	if (bsx_phases()) fprintf(stderr, "[M::msw_plan] pairs %d..%d: %u alignments planned and run on the device\n", p0, p1, nj);
	return BSX_OK;
}

// ------------------------------------------------------------------------------------------
// K6
// ------------------------------------------------------------------------------------------
static int lane_global_batch(bsx_device_t *d, int lane, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res,
                             uint32_t *cigar_pool, size_t cigar_pool_len, bsx_glb_tag_t *tags = nullptr, char **md = nullptr, int64_t *md_cap = nullptr)
{
	if (!d || !d->has_index) return BSX_E_NODEVICE;
	Lane &L = d->lane[lane];
	if (n == 0) return BSX_OK;
	std::lock_guard<std::mutex> hi_lock(L.hi_mu);
	HIPCHK(hipSetDevice(d->ordinal));
	// the last class: queries of any length with their rows in HBM (as lane_extend_batch); a band above 2048 columns stays out of reach
	static const int QCAP[4] = {256, 1024, 16384, 0x7fffffff}, BAND[4] = {256, 1024, 2048, 2048}, NCS[4] = {4, 16, 32, 32}, WPB[4] = {4, 4, 1, 1};
	std::vector<int> order[4];
	size_t zmax[4] = {64, 64, 64, 64};
	int hbm_qmax = 0;
	// tags: the target bases of a job are staged in LDS for the MD walk when they fit next to the DP rows (the rare longer ones are
	// read from HBM); an MD string is at most two characters per target base plus the last count
	static const int TCAP[4] = {512, 2048, 0, 0};
	size_t md_bound = 64;
	const DevScoring &sc = L.sc;
	for (int64_t i = 0; i < n; ++i) {
		const bsx_glb_job_t &j = jobs[i];
		if (tags && j.want_cigar) md_bound += 2 * (size_t)j.tlen + 16;
		if (j.qlen <= 0 || j.tlen <= 0 || j.n_try < 1) { fprintf(stderr, "[bsx-hip] global job %lld: invalid\n", (long long)i); return BSX_E_ARG; }
		if (j.want_cigar && (size_t)j.cigar_off + j.cigar_cap > cigar_pool_len) return BSX_E_ARG;
		const int8_t *mat = j.use_ct ? sc.ctmat : sc.gamat;
		long long wtop = (long long)j.w0 << (j.n_try - 1);
		if (wtop > j.w_max) wtop = j.w_max;
		int dl = j.tlen - j.qlen; dl = dl < 0 ? -dl : dl;
		int max_ins = (int)((double)(((j.qlen + 1) >> 1) * mat[0] - sc.o_ins) / sc.e_ins + 1.);
		int max_del = (int)((double)(((j.qlen + 1) >> 1) * mat[0] - sc.o_del) / sc.e_del + 1.);
		int max_gap = std::max(std::max(max_ins, max_del), 1);
		long long wk = std::max<long long>(std::min<long long>((max_gap + dl + 1) >> 1, wtop), dl + 3);
		long long band = std::min<long long>(j.qlen, 2 * wk + 1);
		int c = 0;
		while (c < 4 && (j.qlen > QCAP[c] || band > BAND[c])) ++c;
		if (c == 4) { fprintf(stderr, "[bsx-hip] global job %lld: a band of %lld columns (query %d, target %d) is beyond the kernel's 2048\n", (long long)i, band, j.qlen, j.tlen); return BSX_E_ARG; }
		if (c == 3) hbm_qmax = std::max(hbm_qmax, j.qlen);
		order[c].push_back((int)i);
		if (j.want_cigar) zmax[c] = std::max(zmax[c], (size_t)band * (size_t)j.tlen + 64);
	}
	int rc, blocks[4];
	size_t ztot = 0;
	for (int c = 0; c < 4; ++c) {
		const long long m = (long long)order[c].size();
		blocks[c] = (int)std::min<long long>((m + WPB[c] - 1) / WPB[c], (long long)d->n_cu * 8);
		zmax[c] = (zmax[c] + 255) & ~(size_t)255;
		// every wave's traceback scratch is sized for the class's largest job (a kilobase read across a long deletion: megabytes): fewer
		// waves rather than tens of GB
		const size_t zbudget = (size_t)4 << 30;
		if ((size_t)blocks[c] * WPB[c] * zmax[c] > zbudget) blocks[c] = (int)std::max<size_t>(1, zbudget / (WPB[c] * zmax[c]));
		if (c == 3 && hbm_qmax) blocks[c] = (int)std::max<size_t>(1, std::min<size_t>((size_t)blocks[c], ((size_t)2 << 30) / global_hbm_row_bytes(hbm_qmax)));
		ztot = std::max(ztot, (size_t)blocks[c] * WPB[c] * zmax[c]);
	}
	if ((rc = L.jobs.reserve((size_t)n * sizeof(bsx_glb_job_t))) != BSX_OK) return rc;
	if ((rc = L.res.reserve((size_t)n * sizeof(bsx_glb_res_t))) != BSX_OK) return rc;
	if ((rc = L.aux.reserve((size_t)n * 4 + 64)) != BSX_OK) return rc;
	// traceback slabs first, then (queries beyond LDS) the DP rows of the HBM class
	const size_t rows_off = (ztot + 256 + 255) & ~(size_t)255, rows_bytes = hbm_qmax ? (size_t)blocks[3] * global_hbm_row_bytes(hbm_qmax) : 0;
	if ((rc = L.scratch.reserve(rows_off + rows_bytes)) != BSX_OK) return rc;
	if ((rc = L.pool.reserve(cigar_pool_len * 4 + 64)) != BSX_OK) return rc;
	unsigned long long *md_cursor = dev_counters(L) + 60;
	if (tags) {
		if ((rc = L.tags.reserve((size_t)n * sizeof(bsx_glb_tag_t))) != BSX_OK) return rc;
		if ((rc = L.mdpool.reserve(md_bound)) != BSX_OK) return rc;
		HIPCHK(hipMemsetAsync(md_cursor, 0, 8, L.st_hi));
		HIPCHK(hipMemsetAsync(L.tags.p, 0xff, (size_t)n * sizeof(bsx_glb_tag_t), L.st_hi));   // l_md = -1: jobs that ask for no CIGAR
	}
	H2D(L.st_hi, L.jobs.p, jobs, (size_t)n * sizeof(bsx_glb_job_t));
	size_t off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		H2D(L.st_hi, (int*)L.aux.p + off, order[c].data(), order[c].size() * 4);
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev0, L.st_hi));
	off = 0;
	for (int c = 0; c < 4; ++c) if (!order[c].empty()) {
		if (c == 3)
			launch_global_hbm(L.st_hi, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_glb_job_t*)L.jobs.p, (const int*)L.aux.p + off,
			                  (long long)order[c].size(), (bsx_glb_res_t*)L.res.p, (uint32_t*)L.pool.p, (uint8_t*)L.scratch.p, zmax[c], hbm_qmax, blocks[c],
			                  tags ? (bsx_glb_tag_t*)L.tags.p : nullptr, (char*)L.mdpool.p, (unsigned long long)md_bound, md_cursor, (char*)L.scratch.p + rows_off);
		else
		launch_global(L.st_hi, d->ix, L.sc, (const uint8_t*)L.reads.p, (const bsx_glb_job_t*)L.jobs.p, (const int*)L.aux.p + off,
		              (long long)order[c].size(), (bsx_glb_res_t*)L.res.p, (uint32_t*)L.pool.p, (uint8_t*)L.scratch.p, zmax[c],
		              QCAP[c], NCS[c], blocks[c], WPB[c], tags ? (bsx_glb_tag_t*)L.tags.p : nullptr, (char*)L.mdpool.p, (unsigned long long)md_bound, md_cursor, tags ? TCAP[c] : 0);
		off += order[c].size();
	}
	HIPCHK(hipEventRecord(L.ev1, L.st_hi));
	if ((rc = finish_timed(L, 4)) != BSX_OK) return rc;
	D2H(L.st_hi, res, L.res.p, (size_t)n * sizeof(bsx_glb_res_t));
	D2H(L.st_hi, cigar_pool, L.pool.p, cigar_pool_len * 4);
	if (tags) {
		unsigned long long used = 0;
		D2H(L.st_hi, tags, L.tags.p, (size_t)n * sizeof(bsx_glb_tag_t));
		D2H(L.st_hi, &used, md_cursor, 8);
		if (used > md_bound) { fprintf(stderr, "[bsx-hip] MD pool overrun (%llu > %zu)\n", used, md_bound); return BSX_E_INTERNAL; }
		if ((int64_t)used + 1 > *md_cap) {
			char *g = (char*)realloc(*md, (size_t)used + 64);
			if (!g) return BSX_E_NOMEM;
			*md = g; *md_cap = (int64_t)used + 64;
		}
		D2H(L.st_hi, *md, L.mdpool.p, (size_t)used);
	}
	return BSX_OK;
}

// the C ABI of include/bsx.h: lane 0
extern "C" BSX_API int bsx_seed_batch(bsx_device_t *d, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks, bsx_intv_t **out, int64_t *out_cap, int64_t *out_off)
{ return lane_seed_batch(d, 0, opt, n, tasks, out, out_cap, out_off); }
extern "C" BSX_API int bsx_regions_batch(bsx_device_t *d, const bsx_opt_t *opt, int64_t n, const bsx_seed_task_t *tasks, bsx_region_t **out, int64_t *out_cap,
                                         int64_t *out_off, int32_t *out_n, bsx_intv_t **decl_intv, int64_t *decl_cap, int64_t *decl_off)
{ return lane_regions_batch(d, 0, opt, n, tasks, out, out_cap, out_off, out_n, decl_intv, decl_cap, decl_off); }
extern "C" BSX_API int bsx_regions_finish(bsx_device_t *d, bsx_region_t **out, int64_t *out_cap, int64_t *out_off, int32_t *out_n)
{ return lane_regions_finish(d, 0, out, out_cap, out_off, out_n); }
extern "C" BSX_API int bsx_regions_dedup_cap(void) { return dedup_cap(); }
extern "C" BSX_API int bsx_regions_dedup(bsx_device_t *d, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx)
{ return lane_regions_dedup(d, 0, opt, n_reads, per_read, out_n, out_idx); }
extern "C" BSX_API int bsx_regions_dedup_long_cap(void) { return dedup_long_cap(); }
extern "C" BSX_API int bsx_regions_dedup2(bsx_device_t *d, const bsx_opt_t *opt, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx,
                                          int64_t *long_off, uint16_t **long_idx, int64_t *long_cap)
{ return lane_regions_dedup(d, 0, opt, n_reads, per_read, out_n, out_idx, long_off, long_idx, long_cap); }
extern "C" BSX_API int bsx_sa_batch(bsx_device_t *d, int64_t n, const bsx_sa_job_t *jobs, uint64_t *pos) { return lane_sa_batch(d, 0, n, jobs, pos); }
extern "C" BSX_API int bsx_extend_batch(bsx_device_t *d, int64_t n, const bsx_ext_job_t *jobs, bsx_ext_res_t *res) { return lane_extend_batch(d, 0, n, jobs, res); }
extern "C" BSX_API int bsx_sw_batch(bsx_device_t *d, int64_t n, const bsx_sw_job_t *jobs, bsx_sw_res_t *res) { return lane_sw_batch(d, 0, n, jobs, res); }
extern "C" BSX_API int bsx_global_batch(bsx_device_t *d, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_len)
{ return lane_global_batch(d, 0, n, jobs, res, cigar_pool, cigar_pool_len); }
extern "C" BSX_API int bsx_global_batch_tags(bsx_device_t *d, int64_t n, const bsx_glb_job_t *jobs, bsx_glb_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_len,
                                             bsx_glb_tag_t *tags, char **md, int64_t *md_cap)
{ if (!tags || !md || !md_cap) return BSX_E_ARG; return lane_global_batch(d, 0, n, jobs, res, cigar_pool, cigar_pool_len, tags, md, md_cap); }

// the seams as one vtable for the host pipeline; ctx = (device, lane)
#define LR(c) ((LaneRef*)(c))->d, ((LaneRef*)(c))->lane
static int be_set_opt(void *c, const bsx_opt_t *o) { return lane_set_opt(LR(c), o); }
static int be_set_reads(void *c, const uint8_t *b, size_t n) { return lane_set_reads(LR(c), b, n); }
static int be_seed(void *c, const bsx_opt_t *o, int64_t n, const bsx_seed_task_t *t, bsx_intv_t **out, int64_t *cap, int64_t *off) { return lane_seed_batch(LR(c), o, n, t, out, cap, off); }
static int be_sa(void *c, int64_t n, const bsx_sa_job_t *j, uint64_t *p) { return lane_sa_batch(LR(c), n, j, p); }
static int be_ext(void *c, int64_t n, const bsx_ext_job_t *j, bsx_ext_res_t *r) { return lane_extend_batch(LR(c), n, j, r); }
static int be_sw(void *c, int64_t n, const bsx_sw_job_t *j, bsx_sw_res_t *r) { return lane_sw_batch(LR(c), n, j, r); }
static int be_regions(void *c, const bsx_opt_t *o, int64_t n, const bsx_seed_task_t *t, bsx_region_t **out, int64_t *cap, int64_t *off, int32_t *cnt,
                      bsx_intv_t **di, int64_t *dc, int64_t *doff)
{
	// The handful of strand searches seeded again (tandem repeats) are done by the time the region tiers are, so by default they are
	// collected before returning; $BSX_ASYNC_REDO=1 leaves them pending for regions_finish at the start of the chunk's back half.
	const int async_redo = (int)bsx_tune_long("async_redo", 0);
	int rc = lane_regions_batch(LR(c), o, n, t, out, cap, off, cnt, di, dc, doff);
	if (rc == BSX_OK && !async_redo) rc = lane_regions_finish(LR(c), out, cap, off, cnt);
	return rc;
}
static int be_regions_finish(void *c, bsx_region_t **out, int64_t *cap, int64_t *off, int32_t *cnt) { return lane_regions_finish(LR(c), out, cap, off, cnt); }
static int be_dedup(void *c, const bsx_opt_t *o, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx) { return lane_regions_dedup(LR(c), o, n_reads, per_read, out_n, out_idx); }
static int be_msw_plan(void *c, const bsx_opt_t *o, const bsx_pestat_t *pes, int64_t token, int64_t n_reads, int per_read, const uint32_t *roff, int max_len,
                       int p0, int p1, void *table, bsx_sw_res_t **res, int64_t *res_cap, int64_t *n_jobs)
{ return lane_msw_plan(LR(c), o, pes, token, n_reads, per_read, roff, max_len, p0, p1, table, res, res_cap, n_jobs); }
static int be_dedup2(void *c, const bsx_opt_t *o, int64_t n_reads, int per_read, int32_t *out_n, uint8_t *out_idx, int64_t *long_off, uint16_t **long_idx, int64_t *long_cap)
{ return lane_regions_dedup(LR(c), o, n_reads, per_read, out_n, out_idx, long_off, long_idx, long_cap); }
static int be_glb(void *c, int64_t n, const bsx_glb_job_t *j, bsx_glb_res_t *r, uint32_t *pool, size_t len) { return lane_global_batch(LR(c), n, j, r, pool, len); }
static int be_glb_tags(void *c, int64_t n, const bsx_glb_job_t *j, bsx_glb_res_t *r, uint32_t *pool, size_t len, bsx_glb_tag_t *t, char **md, int64_t *cap)
{ return lane_global_batch(LR(c), n, j, r, pool, len, t, md, cap); }

static LaneRef g_lane_ref[8][BSX_LANES];   // ctx storage for the vtables (by device ordinal)

extern "C" int bsx_hip_backend_lane(bsx_device_t *dev, int lane, bsx_backend_t *out)
{
	if (!dev) return BSX_E_NODEVICE;
	if (lane < 0 || lane >= BSX_LANES || dev->ordinal < 0 || dev->ordinal >= 8) return BSX_E_ARG;
	LaneRef *r = &g_lane_ref[dev->ordinal][lane];
	r->d = dev; r->lane = lane;
	memset(out, 0, sizeof(*out));
	out->ctx = r; out->name = "hip-gfx950";
	out->set_opt = be_set_opt; out->set_reads = be_set_reads; out->seed_batch = be_seed; out->sa_batch = be_sa;
	out->extend_batch = be_ext; out->sw_batch = be_sw; out->global_batch = be_glb; out->global_batch_tags = be_glb_tags;
	out->regions_batch = bsx_tune_long("host_chain", 0) ? nullptr : be_regions;
	out->regions_finish = out->regions_batch ? be_regions_finish : nullptr;   // BSX_HOST_CHAIN=1: host chaining for every task (A/B checks)
	out->regions_dedup = out->regions_batch && !bsx_tune_long("host_dedup", 0) ? be_dedup : nullptr;   // BSX_HOST_DEDUP=1: C5 on the host for every read (A/B checks)
	out->dedup_cap = dedup_cap();
	out->regions_dedup2 = out->regions_dedup ? be_dedup2 : nullptr;
	out->msw_plan = out->regions_dedup ? be_msw_plan : nullptr;
	return BSX_OK;
}
extern "C" int bsx_hip_backend(bsx_device_t *dev, bsx_backend_t *out) { return bsx_hip_backend_lane(dev, 0, out); }
