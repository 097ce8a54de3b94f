/* tune.c -- the registry behind tune.h: named settings, $BSX_TUNE parsed once, bsx_tune_set() at run time. */
#include <pthread.h>
#include "bsx_core.h"
#include "tune.h"

typedef struct { const char *name, *doc; char *val; } knob_t;
static knob_t g_knobs[] = {
	/* what the tests and the A/B tools switch; defaults in brackets */
	{"phases", "[0] 1: per-phase lines on stderr and the kernels' cycle counters; 2: the stage counters read after every region launch (also $BSX_PHASES)", 0},
	{"tiers", "[0] 1: HIP events between the region launches of a chunk, printed (no cycle counters): bench.py's stand-alone chunk", 0},
	{"seed_form", "[table] classic: seeding without the table of k-mer intervals (k_seed.hip: the reference's own sequence of bwt_extend calls, what counts its FM-block touches)", 0},
	{"seed_tab_k", "[from the text's size: 18 at 3.1 Gbp] levels of the table of k-mer intervals; 0: none", 0},
	{"seed_mem_cap", "[max(64, longest read)] entries of a strand search's first-pass interval list (tests: short lists, so that ordinary reads are seeded again)", 0},
	{"seed_direct", "[1] 0: interval lists copied behind each other instead of written where they stay", 0},
	{"seed_quota", "[0 = persistent lanes; 1 for seed_form=classic] strand searches a lane of the seeding kernel takes", 0},
	{"tier3_early", "[1] the last HBM tier's launch for what is known to need it when the occurrences are counted: on a stream of its own beside the other tiers; 0: behind them", 0},
	{"tier3_order", "[1] the last HBM tier takes its strand searches longest first (by occurrences to visit); 0: in the order the tier before handed them on", 0},
	{"tier3_wgs", "[2] workgroups of four waves per CU in the last HBM tier's launch (1..3)", 0},
	{"seed_budget2", "[8] the second seeding pass's budget in first-pass budgets; what exceeds it is seeded a third time on the side stream, without one", 0},
	{"seed_trip_budget", "[4096] FM extensions (per 256 bases of read) after which the first seeding pass hands a strand search to the second; 0: never", 0},
	{"redo_merge_min", "[4096] from how many overflowed strand searches the second seeding pass runs inside the chunk's launch sequence (1: always; negative or huge: never)", 0},
	{"async_redo", "[0] 1: bsx_process_seqs collects the strand searches seeded again on the side stream only in the back half", 0},
	{"device_sa_intv", "[2] the device's suffix-array sample: every n-th rank (a power of two up to the files' 32)", 0},
	{"pos_cap", "[384 per strand search] seed occurrences the chunk-wide position table holds (tests: strand searches that find no room)", 0},
	{"ssw_cap", "[24 per strand search] jobs the seed filter's chunk-wide list holds (tests: seeds that find no room)", 0},
	{"regions_quota", "[16] strand searches per wave of the first region tier", 0},
	{"mid_quota", "[8] strand searches per wave of the larger LDS tiers", 0},
	{"c2r_quota", "[16] strand searches per wave of the chains -> regions launch", 0},
	{"regions_occ", "[5] waves per SIMD the first region tier's register allocation targets (3..5)", 0},
	{"regions_mid", "[1] 0: no second LDS tier", 0},
	{"tier1c", "[1] 0: no third LDS tier / second chains -> regions launch for ordinary reads (the tier sequence of rounds 2-4)", 0},
	{"x4", "[1] 0: every extension inline in the chains -> regions loop (no k_x4prep / k_extl / k_ext4 ahead of it)", 0},
	{"xl", "[1] 0: the narrow extension jobs through k_ext4 as well (no lane-per-job kernel)", 0},
	{"ext_win", "[1] chunks with long reads: an extension's rows in five register slots that follow the band (ext_dp_win); 0: in LDS (the form of rounds 3-5)", 0},
	{"ext4", "[off] tests: bsx_extend_batch through the quarter-wave kernel (1) / then the lane-per-job kernel (2) of the regions path / through the wavefront-per-job form with a register window (3)", 0},
	{"chain_stages", "[3] how consecutive chunks' front halves are chained on the device: 0 none, 1 seeding, 2 seeding and regions, 3 the same but the HBM tiers hold nobody back, 4 strictly one after the other", 0},
	{"small_copies_unmasked", "[1] the front half's copies of less than 256 KB go through the lane's unmasked stream (the CUs the front-half streams leave alone); 0: through the stream they are ordered on", 0},
	{"reserve_cu_every", "[0] n >= 2: the front-half streams leave one compute unit in n (of every shader engine of every XCD) to the back half's short batches; < 2: none", 0},
	{"host_chain", "[0] 1: every strand search chained on the host over the batch kernels (A/B against the region kernels)", 0},
	{"host_dedup", "[0] 1: mem_sort_deduplicate of every read on the host (A/B against k_dedup)", 0},
	{"stream_whole_chunk", "[0] N: a chunk's own thread runs its back half too, at most N at a time", 0},
	{"no_chunk_scan", "[0] 1 (several ranks): no boundary scan of plain input files, every rank parses everything", 0},
	{"no_chunk_skip", "[0] 1 (several ranks): the other ranks' chunks are parsed into records and dropped instead of walked", 0},
	{"index_batch", "[256 M] suffixes per batch of the device index builder's first round", 0},
	{"back_slices", "[4] slices the pairs of a chunk are cut into for the stages after the insert-size statistics (mate rescue .. SAM text); 1: the stages over the whole chunk", 0},
	{"back_threads", "[2] threads that take those slices alternately (one waits for its slice's K5 / K6 batch while the other runs host stages); up to 4", 0},
	{"shard_pairs", "[0] several ranks (RANK / WORLD_SIZE): 1 = every rank aligns its slice of EVERY chunk (inputs with fewer chunks than GPUs) instead of every world-th chunk", 0},
	{"gather_via_rank0", "[0] several ranks with $BSX_OUT: 1 = the records go through rank 0 even where every rank could write its own chunks into the file", 0},
	{"gather_transport", "[rccl] several ranks: socket = Unix-domain sockets through rank 0 instead of RCCL", 0},
	{"tier2_export", "[0] the first HBM tier of the region kernels in steps -- chains exported, seeds extended ahead (k_extl / k_ext4; 1: every chain's best seed, 2: every seed of every main list), the seed loop by a chains -> regions launch with its regions in HBM -- instead of its monolithic form (same SAM; measured no faster, DESIGN.md section 4)", 0},
	{"msw_plan", "[0] 1: mate rescue's plan pass (which candidates need an alignment, over which window) by k_msw_plan over the lists on the device, its K5 batch run from device memory, instead of by the host's first replay pass (same SAM; measured slower, DESIGN.md section 4)", 0},
	{"long_dedup", "[1] 0: reads with more than 32 regions are de-duplicated on the host (A/B against k_dedup_long)", 0},
};
#define N_KNOBS ((int)(sizeof(g_knobs) / sizeof(g_knobs[0])))
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_env_done = 0, g_phases_env = -1;

static int find(const char *name)
{
	int i;
	for (i = 0; i < N_KNOBS; ++i) if (strcmp(g_knobs[i].name, name) == 0) return i;
	return -1;
}

static void set_locked(int i, const char *value, size_t len)
{
	free(g_knobs[i].val);
	g_knobs[i].val = 0;
	if (value) { g_knobs[i].val = (char*)malloc(len + 1); memcpy(g_knobs[i].val, value, len); g_knobs[i].val[len] = 0; }
}

/* $BSX_TUNE, once: "name=value,name=value" (a name without '=' means 1) */
static void env_locked(void)
{
	const char *e, *p;
	if (g_env_done) return;
	g_env_done = 1;
	e = getenv("BSX_TUNE");
	for (p = e; p && *p; ) {
		const char *q = strchr(p, ','), *eq;
		size_t n = q ? (size_t)(q - p) : strlen(p);
		char name[64];
		eq = (const char*)memchr(p, '=', n);
		if (n && (eq ? (size_t)(eq - p) : n) < sizeof(name)) {
			size_t ln = eq ? (size_t)(eq - p) : n;
			int i;
			memcpy(name, p, ln); name[ln] = 0;
			i = find(name);
			if (i < 0) fprintf(stderr, "[W::bsx_tune] $BSX_TUNE: no setting named \"%s\"\n", name);
			else if (eq) set_locked(i, eq + 1, n - ln - 1);
			else set_locked(i, "1", 1);
		}
		p = q ? q + 1 : 0;
	}
}

BSX_API int bsx_tune_set(const char *name, const char *value)
{
	int i;
	if (!name || (i = find(name)) < 0) return BSX_E_ARG;
	pthread_mutex_lock(&g_mu);
	env_locked();
	set_locked(i, value, value ? strlen(value) : 0);
	pthread_mutex_unlock(&g_mu);
	return BSX_OK;
}

/* (the returned text stays valid until the same name is set again: settings change between calls of the library, not during them) */
BSX_API const char *bsx_tune_str(const char *name)
{
	const char *v = 0;
	int i = find(name);
	if (i < 0) { fprintf(stderr, "[E::bsx_tune] the library asks for a setting that is not in its table: \"%s\"\n", name); return 0; }
	pthread_mutex_lock(&g_mu);
	env_locked();
	v = g_knobs[i].val;
	pthread_mutex_unlock(&g_mu);
	return v;
}
BSX_API long bsx_tune_long(const char *name, long dflt) { const char *v = bsx_tune_str(name); return v && *v ? strtol(v, 0, 10) : dflt; }
BSX_API int bsx_tune_is_set(const char *name) { return bsx_tune_str(name) != 0; }

BSX_API int bsx_phases(void)
{
	const char *v = bsx_tune_str("phases");
	if (v) return atoi(v);
	if (g_phases_env < 0) { const char *e = getenv("BSX_PHASES"); g_phases_env = e ? (atoi(e) > 0 ? atoi(e) : 1) : 0; }   /* (any value, "" included, turns it on, as before) */
	return g_phases_env;
}

BSX_API const char *bsx_tune_name(int i) { return i >= 0 && i < N_KNOBS ? g_knobs[i].name : 0; }
BSX_API const char *bsx_tune_doc(int i) { return i >= 0 && i < N_KNOBS ? g_knobs[i].doc : 0; }
