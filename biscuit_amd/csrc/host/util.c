/* util.c -- ordering primitives whose exact behaviour leaks into the SAM output.
 *
 * The reference sorts with klib's non-stable ks_introsort (lib/aln/ksort.h:184-236) and chains
 * seeds through a klib B-tree (lib/aln/kbtree.h).  The permutation an unstable sort produces for
 * equal keys, and which of several equal keys a B-tree lookup lands on, decide which chain or
 * region is processed first.  Both are re-implemented here from their algorithmic description
 * (index based, element-width generic) so that the comparison/swap sequence -- and therefore the
 * permutation -- is the same.  Pinned against the reference templates in tests/test_host_primitives.py.
 */
#include <pthread.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <malloc.h>
#include "bsx_core.h"
#include "sort_tmpl.h"

BSX_API int bsx_verbose = 3;

/* hash_64 (lib/aln/utils.h:107-117) */
uint64_t bsx_hash64(uint64_t key)
{
	key += ~(key << 32);
	key ^= (key >> 22);
	key += ~(key << 13);
	key ^= (key >> 8);
	key += (key << 3);
	key ^= (key >> 15);
	key += ~(key << 27);
	key ^= (key >> 31);
	return key;
}

/* ------------------------------------------------------------------------------------------
 * introsort: median-of-3 quicksort that leaves runs of <=16 for one final insertion sort,
 * with a comb-sort fallback when the depth budget 2*ceil(log2 n) is spent.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	char *base; size_t w; bsx_lt_fn lt; char *tmp, *pivot;
} srt_t;

#define EL(S, i) ((S)->base + (size_t)(i) * (S)->w)

static inline void srt_swap(srt_t *S, ptrdiff_t i, ptrdiff_t j)
{
	memcpy(S->tmp, EL(S, i), S->w); memcpy(EL(S, i), EL(S, j), S->w); memcpy(EL(S, j), S->tmp, S->w);
}

static void srt_insertion(srt_t *S, ptrdiff_t s, ptrdiff_t t) /* [s, t) */
{
	ptrdiff_t i, j;
	for (i = s + 1; i < t; ++i)
		for (j = i; j > s && S->lt(EL(S, j), EL(S, j - 1)); --j)
			srt_swap(S, j, j - 1);
}

static void srt_comb(srt_t *S, ptrdiff_t s, size_t n) /* ksort.h:162-183 */
{
	const double shrink_factor = 1.2473309501039786540366528676643;
	int swapped;
	size_t gap = n, i;
	do {
		if (gap > 2) {
			gap = (size_t)(gap / shrink_factor);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = 0;
		for (i = 0; i + gap < n; ++i)
			if (S->lt(EL(S, s + i + gap), EL(S, s + i))) { srt_swap(S, s + i, s + i + gap); swapped = 1; }
	} while (swapped || gap > 2);
	if (gap != 1) srt_insertion(S, s, s + n);
}

void bsx_introsort(void *base, size_t n, size_t width, bsx_lt_fn lt)
{
	srt_t S;
	struct frame { ptrdiff_t left, right; int depth; } *stack, *top;
	ptrdiff_t s, t, i, j, k;
	int d;
	char buf[512];

	if (n < 1) return;
	S.base = (char*)base; S.w = width; S.lt = lt;
	S.tmp = width * 2 <= sizeof(buf) ? buf : (char*)malloc(width * 2);
	S.pivot = S.tmp + width;
	if (n == 2) {
		if (lt(EL(&S, 1), EL(&S, 0))) srt_swap(&S, 0, 1);
		goto done;
	}
	for (d = 2; 1ul << d < n; ++d);
	stack = (struct frame*)malloc(sizeof(*stack) * (sizeof(size_t) * d + 2));
	top = stack; s = 0; t = (ptrdiff_t)n - 1; d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				srt_comb(&S, s, (size_t)(t - s + 1));
				t = s;
				continue;
			}
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(EL(&S, k), EL(&S, i))) {
				if (lt(EL(&S, k), EL(&S, j))) k = j;
			} else k = lt(EL(&S, j), EL(&S, i)) ? i : j;
			memcpy(S.pivot, EL(&S, k), width);
			if (k != t) srt_swap(&S, k, t);
			for (;;) {
				do ++i; while (lt(EL(&S, i), S.pivot));
				do --j; while (i <= j && lt(S.pivot, EL(&S, j)));
				if (j <= i) break;
				srt_swap(&S, i, j);
			}
			srt_swap(&S, i, t);
			if (i - s > t - i) {
				if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == stack) {
				free(stack);
				srt_insertion(&S, 0, (ptrdiff_t)n);
				break;
			}
			--top; s = top->left; t = top->right; d = top->depth;
		}
	}
done:
	if (S.tmp != buf) free(S.tmp);
}

#define LT_VAL(a, b) (*(a) < *(b))
BSX_SORT_DEFINE(sort_u64, uint64_t, LT_VAL)
BSX_SORT_DEFINE(sort_i64, int64_t, LT_VAL)
void bsx_introsort_u64(size_t n, uint64_t *a) { sort_u64(n, a); }
void bsx_introsort_i64(size_t n, int64_t *a) { sort_i64(n, a); }

/* ------------------------------------------------------------------------------------------
 * B-tree, minimum degree 3 (the value kbtree.h:75 derives for 72-byte keys in 512-byte nodes),
 * pre-emptive split on the way down, lower-bound search inside a node, duplicates allowed.
 * ------------------------------------------------------------------------------------------ */
#define BT_T 3
#define BT_MAXK (2 * BT_T - 1)

typedef struct {
	int32_t is_internal, n;
	int64_t key[BT_MAXK];
	int32_t id[BT_MAXK];
	int32_t child[BT_MAXK + 1];
} bt_node_t;

struct bsx_btree {
	bt_node_t *nodes;
	int32_t n_nodes, m_nodes, root, n_keys;
};

static int32_t bt_alloc(bsx_btree_t *t, int internal)
{
	bt_node_t *x;
	if (t->n_nodes == t->m_nodes) {
		t->m_nodes = t->m_nodes ? t->m_nodes << 1 : 16;
		t->nodes = (bt_node_t*)realloc(t->nodes, sizeof(bt_node_t) * t->m_nodes);
	}
	x = &t->nodes[t->n_nodes];
	memset(x, 0, sizeof(*x));
	x->is_internal = internal;
	return t->n_nodes++;
}

bsx_btree_t *bsx_bt_new(void)
{
	bsx_btree_t *t = (bsx_btree_t*)calloc(1, sizeof(*t));
	bsx_bt_clear(t);
	return t;
}
void bsx_bt_clear(bsx_btree_t *t) { t->n_nodes = 0; t->n_keys = 0; t->root = bt_alloc(t, 0); }
void bsx_bt_free(bsx_btree_t *t) { if (t) { free(t->nodes); free(t); } }
int bsx_bt_size(const bsx_btree_t *t) { return t->n_keys; }

/* position of the last key <= pos inside a node, the first equal key when there are several;
 * *r = sign(pos - key[found]) as in __kb_getp_aux (kbtree.h:118-131) */
static int bt_find(const bt_node_t *x, int64_t pos, int *r)
{
	int begin = 0, end = x->n, rr;
	if (x->n == 0) return -1;
	while (begin < end) {
		int mid = (begin + end) >> 1;
		if (x->key[mid] < pos) begin = mid + 1;
		else end = mid;
	}
	if (begin == x->n) { if (r) *r = 1; return x->n - 1; }
	rr = (x->key[begin] < pos) - (pos < x->key[begin]);
	if (r) *r = rr;
	if (rr < 0) --begin;
	return begin;
}

static void bt_split(bsx_btree_t *t, int32_t xi, int i, int32_t yi)
{
	int32_t zi = bt_alloc(t, t->nodes[yi].is_internal);
	bt_node_t *x = &t->nodes[xi], *y = &t->nodes[yi], *z = &t->nodes[zi];
	z->n = BT_T - 1;
	memcpy(z->key, y->key + BT_T, sizeof(int64_t) * (BT_T - 1));
	memcpy(z->id, y->id + BT_T, sizeof(int32_t) * (BT_T - 1));
	if (y->is_internal) memcpy(z->child, y->child + BT_T, sizeof(int32_t) * BT_T);
	y->n = BT_T - 1;
	memmove(x->child + i + 2, x->child + i + 1, sizeof(int32_t) * (x->n - i));
	x->child[i + 1] = zi;
	memmove(x->key + i + 1, x->key + i, sizeof(int64_t) * (x->n - i));
	memmove(x->id + i + 1, x->id + i, sizeof(int32_t) * (x->n - i));
	x->key[i] = y->key[BT_T - 1];
	x->id[i] = y->id[BT_T - 1];
	++x->n;
}

void bsx_bt_put(bsx_btree_t *t, int64_t pos, int32_t id)
{
	int32_t xi;
	++t->n_keys;
	if (t->nodes[t->root].n == BT_MAXK) {
		int32_t r = t->root, s = bt_alloc(t, 1);
		t->nodes[s].child[0] = r;
		t->root = s;
		bt_split(t, s, 0, r);
	}
	xi = t->root;
	for (;;) {
		bt_node_t *x = &t->nodes[xi];
		int i;
		if (!x->is_internal) {
			i = bt_find(x, pos, 0);
			if (i != x->n - 1) {
				memmove(x->key + i + 2, x->key + i + 1, sizeof(int64_t) * (x->n - i - 1));
				memmove(x->id + i + 2, x->id + i + 1, sizeof(int32_t) * (x->n - i - 1));
			}
			x->key[i + 1] = pos; x->id[i + 1] = id;
			++x->n;
			return;
		}
		i = bt_find(x, pos, 0) + 1;
		if (t->nodes[x->child[i]].n == BT_MAXK) {
			bt_split(t, xi, i, x->child[i]);
			x = &t->nodes[xi]; /* nodes may have been reallocated */
			if (pos > x->key[i]) ++i;
		}
		xi = x->child[i];
	}
}

int32_t bsx_bt_lower(const bsx_btree_t *t, int64_t pos)
{
	int32_t xi = t->root, lower = -1;
	for (;;) {
		const bt_node_t *x = &t->nodes[xi];
		int r = 0, i = bt_find(x, pos, &r);
		if (i >= 0 && r == 0) return x->id[i];
		if (i >= 0) lower = x->id[i];
		if (!x->is_internal) return lower;
		xi = x->child[i + 1];
	}
}

static void bt_walk(const bsx_btree_t *t, int32_t xi, int32_t *ids, int *n)
{
	const bt_node_t *x = &t->nodes[xi];
	int i;
	for (i = 0; i < x->n; ++i) {
		if (x->is_internal) bt_walk(t, x->child[i], ids, n);
		ids[(*n)++] = x->id[i];
	}
	if (x->is_internal) bt_walk(t, x->child[x->n], ids, n);
}
int bsx_bt_traverse(const bsx_btree_t *t, int32_t *ids)
{
	int n = 0;
	bt_walk(t, t->root, ids, &n);
	return n;
}

/* ------------------------------------------------------------------------------------------
 * parallel for
 * ------------------------------------------------------------------------------------------ */
/* Persistent worker pool: the pipeline issues thousands of short parallel loops per chunk (one or
 * two per extension round), so threads are created once and parked on a condition variable. */
typedef struct {
	bsx_for_fn fn; void *data; long n; volatile long next; long grain; int n_part;
	int arena_set;   /* of the calling thread: the workers allocate from the same chunk's arenas */
	int n_inside;    /* workers currently running this loop */
} pf_job_t;

/* Several loops can be open at once (the front half of one chunk and the back half of another each run theirs):
 * idle workers join whichever open loop still has iterations left, so a short loop never queues behind a long one. */
#define PF_MAX_JOBS 4
static struct {
	pthread_mutex_t mu;
	pthread_cond_t cv_work, cv_done;
	int n_workers, m_workers;
	pthread_t *th;
	pf_job_t *jobs[PF_MAX_JOBS];
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, {0, 0, 0, 0} };

/* ---- arenas ---- */
#define ARENA_BLOCK ((size_t)32 << 20)
typedef struct { char *p; size_t cap; } arena_blk_t;
struct bsx_arena { arena_blk_t *blk; int n_blk, m_blk, cur; size_t used; };
BSX_API __thread bsx_arena_t *bsx_tls_arena = 0;
#define ARENA_SETS 4
static bsx_arena_t **g_arenas[ARENA_SETS];
#define ARENA_EXTRA 3
static bsx_arena_t *g_arena_extra[ARENA_SETS][ARENA_EXTRA];   /* for further threads working on the set's chunk beside its owner (the slices of a back half) */
static int g_n_arenas[ARENA_SETS], g_set_busy[ARENA_SETS];
static pthread_mutex_t g_arena_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_arena_cv = PTHREAD_COND_INITIALIZER;
static __thread int tls_arena_set = -1;

/* Large per-chunk arrays (tasks, read buffer, regions...) are kept from chunk to chunk with the arena set: each is hundreds
 * of MB, and taking them from malloc in a worker thread means an mmap, a page fault per 4 KB and a munmap every chunk. */
#define BIG_SLOTS 16
static struct { void *p; size_t cap; } g_big[ARENA_SETS][BIG_SLOTS];
void *bsx_big_get(int set, int slot, size_t bytes)
{
	if (set < 0 || slot < 0 || slot >= BIG_SLOTS) return malloc(bytes);
	if (g_big[set][slot].cap < bytes) {
		free(g_big[set][slot].p);
		g_big[set][slot].cap = bytes + (bytes >> 3) + 4096;
		g_big[set][slot].p = malloc(g_big[set][slot].cap);
	}
	return g_big[set][slot].p;
}
void bsx_big_put(int set, int slot, void *p) { if (set < 0 || slot < 0 || slot >= BIG_SLOTS) free(p); }
/* a realloc'd buffer handed out by bsx_big_get may have been grown by its user: remember the new block */
void bsx_big_update(int set, int slot, void *p, size_t cap) { if (set >= 0 && slot >= 0 && slot < BIG_SLOTS) { g_big[set][slot].p = p; g_big[set][slot].cap = cap; } }

void *bsx_arena_alloc(bsx_arena_t *a, size_t n)
{
	void *q;
	n = (n + 15) & ~(size_t)15;
	while (a->cur < a->n_blk && a->used + n > a->blk[a->cur].cap) { ++a->cur; a->used = 0; }
	if (a->cur == a->n_blk) {
		size_t cap = n > ARENA_BLOCK ? n : ARENA_BLOCK;
		if (a->n_blk == a->m_blk) { a->m_blk = a->m_blk ? a->m_blk << 1 : 8; a->blk = (arena_blk_t*)realloc(a->blk, sizeof(arena_blk_t) * a->m_blk); }
		a->blk[a->n_blk].p = (char*)malloc(cap); a->blk[a->n_blk].cap = cap;
		++a->n_blk; a->used = 0;
	}
	q = a->blk[a->cur].p + a->used;
	a->used += n;
	return q;
}

int bsx_arenas_begin(int n_threads)
{
	int i, set;
	pthread_mutex_lock(&g_arena_mu);
	for (;;) {
		for (set = 0; set < ARENA_SETS; ++set) if (!g_set_busy[set]) break;
		if (set < ARENA_SETS) break;
		pthread_cond_wait(&g_arena_cv, &g_arena_mu);
	}
	g_set_busy[set] = 1;
	if (n_threads > g_n_arenas[set]) {
		g_arenas[set] = (bsx_arena_t**)realloc(g_arenas[set], sizeof(bsx_arena_t*) * n_threads);
		for (i = g_n_arenas[set]; i < n_threads; ++i) g_arenas[set][i] = (bsx_arena_t*)calloc(1, sizeof(bsx_arena_t));
		g_n_arenas[set] = n_threads;
	}
	pthread_mutex_unlock(&g_arena_mu);
	bsx_arenas_bind(set);
	return set;
}

void bsx_arenas_bind(int set)
{
	tls_arena_set = set;
	bsx_tls_arena = set >= 0 ? g_arenas[set][0] : 0;
}

/* a thread that works on the set's chunk beside the one that owns it: arena k of the set's spares is its own (the caller of a parallel loop
 * allocates as thread 0 of the loop, and two callers must not share an arena) */
void bsx_arenas_bind_extra(int set, int k)
{
	if (set < 0 || k < 0 || k >= ARENA_EXTRA) { bsx_arenas_bind(set); return; }
	pthread_mutex_lock(&g_arena_mu);
	if (!g_arena_extra[set][k]) g_arena_extra[set][k] = (bsx_arena_t*)calloc(1, sizeof(bsx_arena_t));
	pthread_mutex_unlock(&g_arena_mu);
	tls_arena_set = set;
	bsx_tls_arena = g_arena_extra[set][k];
}

void bsx_arenas_end(int set)
{
	int i;
	bsx_arenas_bind(-1);
	if (set < 0) return;
	pthread_mutex_lock(&g_arena_mu);
	for (i = 0; i < g_n_arenas[set]; ++i) { g_arenas[set][i]->cur = 0; g_arenas[set][i]->used = 0; }
	for (i = 0; i < ARENA_EXTRA; ++i) if (g_arena_extra[set][i]) { g_arena_extra[set][i]->cur = 0; g_arena_extra[set][i]->used = 0; }
	g_set_busy[set] = 0;
	pthread_cond_signal(&g_arena_cv);
	pthread_mutex_unlock(&g_arena_mu);
}

static void pf_run(pf_job_t *J, int tid)
{
	/* (the caller runs as thread 0 on the arena it is bound to: the owner's, or a helper's own -- bsx_arenas_bind_extra) */
	if (tid > 0) bsx_tls_arena = (J->arena_set >= 0 && tid < g_n_arenas[J->arena_set]) ? g_arenas[J->arena_set][tid] : 0;
	for (;;) {
		long b = __sync_fetch_and_add(&J->next, J->grain), e, i;
		if (b >= J->n) break;
		e = b + J->grain < J->n ? b + J->grain : J->n;
		for (i = b; i < e; ++i) J->fn(J->data, i, tid);
	}
}

static void *pf_worker(void *arg)
{
	int id = (int)(intptr_t)arg;   /* participates as tid id+1 */
	/* the workers yield to the threads that feed the device (the front half of the next chunk, the HIP runtime's
	 * own threads) when there are fewer cores than runnable threads */
	setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 5);
	pthread_mutex_lock(&g_pool.mu);
	for (;;) {
		pf_job_t *J = 0;
		int k;
		for (k = 0; k < PF_MAX_JOBS; ++k) {
			pf_job_t *c = g_pool.jobs[k];
			if (c && id + 1 < c->n_part && c->next < c->n) { J = c; break; }
		}
		if (!J) { pthread_cond_wait(&g_pool.cv_work, &g_pool.mu); continue; }
		++J->n_inside;
		pthread_mutex_unlock(&g_pool.mu);
		pf_run(J, id + 1);
		pthread_mutex_lock(&g_pool.mu);
		if (--J->n_inside == 0) pthread_cond_broadcast(&g_pool.cv_done);
	}
	return 0;
}

void bsx_parallel_for(int n_threads, bsx_for_fn fn, void *data, long n)
{
	long i;
	int slot;
	pf_job_t J;
	if (n <= 0) return;
	if (n_threads > n) n_threads = (int)n;
	if (n_threads <= 1) { for (i = 0; i < n; ++i) fn(data, i, 0); return; }
	pthread_mutex_lock(&g_pool.mu);
	while (g_pool.n_workers < n_threads - 1) { /* grow the pool on demand */
		if (g_pool.n_workers == g_pool.m_workers) {
			g_pool.m_workers = g_pool.m_workers ? g_pool.m_workers << 1 : 16;
			g_pool.th = (pthread_t*)realloc(g_pool.th, sizeof(pthread_t) * g_pool.m_workers);
		}
		if (pthread_create(&g_pool.th[g_pool.n_workers], 0, pf_worker, (void*)(intptr_t)g_pool.n_workers) != 0) break;
		pthread_detach(g_pool.th[g_pool.n_workers]);
		++g_pool.n_workers;
	}
	if (n_threads - 1 > g_pool.n_workers) n_threads = g_pool.n_workers + 1;
	J.fn = fn; J.data = data; J.n = n; J.next = 0; J.n_part = n_threads; J.arena_set = tls_arena_set; J.n_inside = 0;
	J.grain = n / (n_threads * 8L); if (J.grain < 1) J.grain = 1; if (J.grain > 1024) J.grain = 1024;
	for (;;) { /* a free slot (more than PF_MAX_JOBS callers at once: wait for one to finish) */
		for (slot = 0; slot < PF_MAX_JOBS; ++slot) if (!g_pool.jobs[slot]) break;
		if (slot < PF_MAX_JOBS) break;
		pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
	}
	g_pool.jobs[slot] = &J;
	pthread_cond_broadcast(&g_pool.cv_work);
	pthread_mutex_unlock(&g_pool.mu);
	pf_run(&J, 0);
	pthread_mutex_lock(&g_pool.mu);
	g_pool.jobs[slot] = 0;                 /* nobody new joins; wait for the workers still inside */
	while (J.n_inside > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
	pthread_cond_broadcast(&g_pool.cv_done);   /* a caller may be waiting for the slot */
	pthread_mutex_unlock(&g_pool.mu);
}

typedef struct { char *p; size_t bytes; } zero_par_t;
#define ZERO_BLOCK ((size_t)4 << 20)
static void zero_worker(void *data, long b, int tid)
{
	zero_par_t *Z = (zero_par_t*)data;
	size_t at = (size_t)b * ZERO_BLOCK, len = Z->bytes - at < ZERO_BLOCK ? Z->bytes - at : ZERO_BLOCK;
	(void)tid;
	memset(Z->p + at, 0, len);
}
void *bsx_par_calloc(int n_threads, size_t n, size_t size)
{
	zero_par_t Z;
	Z.bytes = (n ? n : 1) * size;
	Z.p = (char*)malloc(Z.bytes);
	if (!Z.p) return 0;
	if (Z.bytes < 4 * ZERO_BLOCK || n_threads < 2) memset(Z.p, 0, Z.bytes);
	else bsx_parallel_for(n_threads, zero_worker, &Z, (long)((Z.bytes + ZERO_BLOCK - 1) / ZERO_BLOCK));
	return Z.p;
}

/* The host stages allocate millions of small records per chunk from many threads.  With glibc's
 * defaults every arena grows and shrinks in small steps, and each step is an mprotect/munmap that
 * takes the process-wide mmap lock against all page faults of the other workers.  Growing in large
 * steps and never trimming keeps the memory of one chunk for the next one. */
__attribute__((constructor)) static void bsx_tune_malloc(void)
{
	mallopt(M_TOP_PAD, 64 << 20);
	mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
}

int bsx_host_threads(const bsx_opt_t *opt)
{
	const char *e = getenv("BSX_HOST_THREADS");
	int n = e ? atoi(e) : (opt ? opt->n_threads : 1);
	return n > 0 ? n : 1;
}

const char *bsx_version(void) { return "biscuit_amd 0.1 (gfx950)"; }

const char *bsx_strerror(int code)
{
	switch (code) {
	case BSX_OK: return "ok";
	case BSX_E_NODEVICE: return "no usable HIP device (the product path has no CPU fallback)";
	case BSX_E_ARG: return "invalid argument";
	case BSX_E_IO: return "I/O error";
	case BSX_E_NOMEM: return "out of memory";
	case BSX_E_FORMAT: return "malformed index or input file";
	case BSX_E_INTERNAL: return "internal error";
	}
	return "unknown error";
}
