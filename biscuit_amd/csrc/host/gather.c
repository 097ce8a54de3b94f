/* gather.c -- SURVEY 8(e)'s "replicas + gather" in C: the per-chunk alignment records of every rank brought together in input order.
 *
 * One process per GPU; chunk k of the input is aligned by rank k % world (or, with several ranks sharing every chunk, slice r of chunk c is
 * "chunk" c * world + r).  The gather runs in rounds; round r carries chunks r * world .. r * world + world - 1, at most one per rank:
 *   1. every rank tells every rank (has_chunk, n_bytes, header bytes) -- an all-gather of three integers: everybody must see the end;
 *   2. via rank 0: every other rank that has a chunk sends exactly its n_bytes to rank 0, which posts the round's receives together (xGMI is
 *      point to point: every sender has its own link into rank 0) and hands the chunks to the sink in input order;
 *      direct (an output FILE all ranks see -- one node, a regular file): nothing but step 1 moves; every rank derives the offset of its own
 *      chunk (header + the bytes of all chunks before it) and writes it there itself with pwrite();
 *   3. the first round in which some rank has no chunk is the last (round-robin dealing: no later chunk exists).
 * Producers (the aligner's writer thread) hand chunks in with bsx_gather_submit(), which blocks once max_pending chunks wait: a rank holds a
 * few chunks of output whatever the input's size, and the transfer of round r overlaps the alignment of the next chunks.
 *
 * The communication layer is a small vtable (bsx_transport_t, include/bsx.h): RCCL over xGMI in the product (csrc/hip/gather_rccl.hip:
 * ncclAllGather / ncclSend / ncclRecv / ncclAllReduce on device staging buffers), and an in-process one -- the ranks as threads of one
 * process -- under which the CPU suite runs the protocol (sizes, offsets, ordering, uneven ends, failures) at world sizes 2-4.
 * biscuit_amd/gather.py is the same protocol over torch.distributed; it stays as the launcher's fallback (--gather python). */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>
#include "bsx_core.h"

typedef struct { int64_t idx; void *buf; size_t n; } g_item_t;
struct bsx_gather {
	bsx_transport_t tr;
	char *direct_path;
	void (*sink)(void *ud, int64_t chunk, const void *buf, size_t n);
	void *sink_ud;
	/* producer -> consumer */
	pthread_mutex_t mu; pthread_cond_t cv_put, cv_get;
	g_item_t *q; int q_n, q_cap, eof, dead;
	g_item_t *stash; int n_stash, m_stash;
	void *header; size_t header_n;
	/* statistics */
	int64_t rounds, bytes_moved, bytes_written;
	int failed;
};

BSX_API int bsx_gather_open(const bsx_transport_t *tr, const char *direct_path, void (*sink)(void *ud, int64_t chunk, const void *buf, size_t n), void *ud,
                            int max_pending, bsx_gather_t **out)
{
	bsx_gather_t *G;
	if (!tr || !out || tr->world < 1 || tr->rank < 0 || tr->rank >= tr->world) return BSX_E_ARG;
	if (tr->world > 1 && (!tr->all_gather || (!direct_path && (!tr->send || !tr->recv_many)))) return BSX_E_ARG;
	G = (bsx_gather_t*)calloc(1, sizeof(*G));
	G->tr = *tr;
	G->direct_path = direct_path && tr->world > 1 ? strdup(direct_path) : 0;
	G->sink = sink; G->sink_ud = ud;
	G->q_cap = max_pending > 0 ? max_pending : 3;
	G->q = (g_item_t*)calloc((size_t)G->q_cap, sizeof(g_item_t));
	pthread_mutex_init(&G->mu, 0); pthread_cond_init(&G->cv_put, 0); pthread_cond_init(&G->cv_get, 0);
	*out = G;
	return BSX_OK;
}

BSX_API int bsx_gather_set_header(bsx_gather_t *G, const void *hdr, size_t n)
{
	if (!G) return BSX_E_ARG;
	pthread_mutex_lock(&G->mu);
	free(G->header);
	G->header = n ? malloc(n) : 0; G->header_n = n;
	if (n) memcpy(G->header, hdr, n);
	pthread_mutex_unlock(&G->mu);
	return BSX_OK;
}

/* buf: malloc'd, the gather's from now on (freed when its round is done).  Blocks while max_pending chunks wait; returns at once when the
 * gather has ended (another rank failed: nobody takes the chunk any more). */
BSX_API int bsx_gather_submit(bsx_gather_t *G, int64_t chunk, void *buf, size_t n)
{
	if (!G) { free(buf); return BSX_E_ARG; }
	pthread_mutex_lock(&G->mu);
	while (G->q_n == G->q_cap && !G->dead) pthread_cond_wait(&G->cv_put, &G->mu);
	if (G->dead) { pthread_mutex_unlock(&G->mu); free(buf); return BSX_E_IO; }
	G->q[G->q_n].idx = chunk; G->q[G->q_n].buf = buf; G->q[G->q_n].n = n; ++G->q_n;
	pthread_cond_signal(&G->cv_get);
	pthread_mutex_unlock(&G->mu);
	return BSX_OK;
}

BSX_API int bsx_gather_close_input(bsx_gather_t *G)
{
	if (!G) return BSX_E_ARG;
	pthread_mutex_lock(&G->mu);
	G->eof = 1;
	pthread_cond_signal(&G->cv_get);
	pthread_mutex_unlock(&G->mu);
	return BSX_OK;
}

/* this rank's chunk `want`: 1 and *it, or 0 when the producer is done */
static int next_own(bsx_gather_t *G, int64_t want, g_item_t *it)
{
	int k;
	for (;;) {
		for (k = 0; k < G->n_stash; ++k) if (G->stash[k].idx == want) { *it = G->stash[k]; G->stash[k] = G->stash[--G->n_stash]; return 1; }
		pthread_mutex_lock(&G->mu);
		while (G->q_n == 0 && !G->eof) pthread_cond_wait(&G->cv_get, &G->mu);
		if (G->q_n == 0) { pthread_mutex_unlock(&G->mu); return 0; }
		*it = G->q[0];
		memmove(G->q, G->q + 1, sizeof(g_item_t) * (size_t)(--G->q_n));
		pthread_cond_signal(&G->cv_put);
		pthread_mutex_unlock(&G->mu);
		if (it->idx == want) return 1;
		/* (producers deliver in order; kept for safety) */
		if (G->n_stash == G->m_stash) { G->m_stash = G->m_stash ? G->m_stash << 1 : 4; G->stash = (g_item_t*)realloc(G->stash, sizeof(g_item_t) * (size_t)G->m_stash); }
		G->stash[G->n_stash++] = *it;
	}
}

static void gather_fail(bsx_gather_t *G, const char *what, int err)
{
	if (!G->failed) fprintf(stderr, "[E::bsx_gather] rank %d: %s%s%s failed: %s\n", G->tr.rank, what, G->direct_path ? " " : "", G->direct_path ? G->direct_path : "", strerror(err));
	G->failed = 1;
}

static int pwrite_all(int fd, const void *buf, size_t n, int64_t off)
{
	size_t done = 0;
	while (done < n) {
		ssize_t w = pwrite(fd, (const char*)buf + done, n - done, (off_t)(off + (int64_t)done));
		if (w < 0) { if (errno == EINTR) continue; return -1; }
		done += (size_t)w;
	}
	return 0;
}

/* the rounds, on the calling thread, until the input has ended on every rank; *n_chunks = chunks seen (the same on all ranks).
 * Returns BSX_OK, BSX_E_IO when this rank could not write (it still took part in every round: the others are inside the same
 * exchanges), or the transport's error (then the gather is dead and producers are released). */
BSX_API int bsx_gather_run(bsx_gather_t *G, int64_t *n_chunks_out)
{
	const int world = G->tr.world, rank = G->tr.rank;
	int64_t r = 0, n_chunks = 0, base = 0, *metas = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)world);
	int fd = -1, rc = BSX_OK, k;
	int *src = (int*)malloc(sizeof(int) * (size_t)world);
	void **bufs = (void**)calloc((size_t)world, sizeof(void*));
	size_t *lens = (size_t*)calloc((size_t)world, sizeof(size_t)), *caps = (size_t*)calloc((size_t)world, sizeof(size_t));
	void **rbuf = (void**)malloc(sizeof(void*) * (size_t)world);
	size_t *rlen = (size_t*)malloc(sizeof(size_t) * (size_t)world);
	for (;;) {
		g_item_t mine;
		const int has = next_own(G, r * world + rank, &mine);
		int64_t me[3];
		int all_have = 1;
		if (!has) { mine.buf = 0; mine.n = 0; mine.idx = -1; }
		if (world == 1) {
			if (!has) break;
			if (G->sink) G->sink(G->sink_ud, r, mine.buf, mine.n);
			free(mine.buf);
			++n_chunks; ++r; ++G->rounds;
			continue;
		}
		me[0] = has; me[1] = (int64_t)mine.n; me[2] = 0;
		if (G->direct_path && r == 0 && rank == 0) { /* the file exists, with its header, before anybody learns the header's length */
			pthread_mutex_lock(&G->mu); me[2] = (int64_t)G->header_n; pthread_mutex_unlock(&G->mu);
			fd = open(G->direct_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
			if (fd < 0) gather_fail(G, "opening", errno);
			else if (me[2] && pwrite_all(fd, G->header, (size_t)me[2], 0) != 0) gather_fail(G, "writing", errno);
		}
		if ((rc = G->tr.all_gather(G->tr.ctx, me, 3, metas)) != BSX_OK) { free(mine.buf); break; }
		for (k = 0; k < world; ++k) if (!metas[3 * k]) all_have = 0;
		if (G->direct_path) {
			int64_t off;
			if (r == 0) {
				base = metas[2];
				if (rank != 0) { fd = open(G->direct_path, O_WRONLY); if (fd < 0) gather_fail(G, "opening", errno); }
			}
			for (k = 0, off = base; k < rank; ++k) if (metas[3 * k]) off += metas[3 * k + 1];
			if (has && mine.n && !G->failed) {
				if (pwrite_all(fd, mine.buf, mine.n, off) != 0) gather_fail(G, "writing", errno);
				else G->bytes_written += (int64_t)mine.n;
			}
			for (k = 0; k < world; ++k) if (metas[3 * k]) { base += metas[3 * k + 1]; ++n_chunks; }
		} else if (rank == 0) {
			int ns = 0;
			for (k = 1; k < world; ++k) {
				const size_t nb = metas[3 * k] ? (size_t)metas[3 * k + 1] : 0;
				lens[k] = nb;
				if (!nb) continue;
				if (caps[k] < nb) { free(bufs[k]); caps[k] = nb + (nb >> 2) + 4096; bufs[k] = malloc(caps[k]); }
				src[ns] = k; rbuf[ns] = bufs[k]; rlen[ns] = nb; ++ns;
			}
			/* this rank's own chunk first: the sink runs while nothing is in flight yet only in the sequential transports; the RCCL one posts
			 * the receives as a group and they progress on its stream */
			if (ns && (rc = G->tr.recv_many(G->tr.ctx, ns, src, rbuf, rlen)) != BSX_OK) { free(mine.buf); break; }
			if (has) { if (G->sink) G->sink(G->sink_ud, r * world, mine.buf, mine.n); ++n_chunks; }
			for (k = 1; k < world; ++k) if (metas[3 * k]) {
				if (G->sink) G->sink(G->sink_ud, r * world + k, bufs[k], lens[k]);
				G->bytes_moved += (int64_t)lens[k];
				++n_chunks;
			}
		} else {
			for (k = 0; k < world; ++k) if (metas[3 * k]) ++n_chunks;
			if (has && mine.n && (rc = G->tr.send(G->tr.ctx, 0, mine.buf, mine.n)) != BSX_OK) { free(mine.buf); break; }
		}
		free(mine.buf);
		++G->rounds; ++r;
		if (!all_have) break;
	}
	if (fd >= 0 && close(fd) != 0) gather_fail(G, "closing", errno);
	pthread_mutex_lock(&G->mu);
	G->dead = 1;   /* late producers (another rank ended the rounds early) are not blocked */
	while (G->q_n) free(G->q[--G->q_n].buf);
	pthread_cond_broadcast(&G->cv_put);
	pthread_mutex_unlock(&G->mu);
	for (k = 0; k < world; ++k) free(bufs[k]);
	free(bufs); free(lens); free(caps); free(rbuf); free(rlen); free(src); free(metas);
	if (n_chunks_out) *n_chunks_out = n_chunks;
	return rc != BSX_OK ? rc : G->failed ? BSX_E_IO : BSX_OK;
}

BSX_API void bsx_gather_stats(const bsx_gather_t *G, int64_t out[3]) { out[0] = G->rounds; out[1] = G->bytes_moved; out[2] = G->bytes_written; }

BSX_API void bsx_gather_free(bsx_gather_t *G)
{
	int k;
	if (!G) return;
	for (k = 0; k < G->q_n; ++k) free(G->q[k].buf);
	for (k = 0; k < G->n_stash; ++k) free(G->stash[k].buf);
	free(G->q); free(G->stash); free(G->header); free(G->direct_path);
	pthread_mutex_destroy(&G->mu); pthread_cond_destroy(&G->cv_put); pthread_cond_destroy(&G->cv_get);
	free(G);
}

/* May every rank write its own chunks into `path`?  Only when all ranks are on one node -- they must see the same file -- and the target is
 * (or will be created as) a regular file: pwrite() at offsets needs a seekable file (gather.py: direct_output_ok). */
BSX_API int bsx_gather_direct_ok(const char *path, int world, int local_world)
{
	struct stat st;
	if (!path || local_world != world) return 0;
	if (stat(path, &st) == 0) return S_ISREG(st.st_mode);
	return errno == ENOENT;
}

/* ------------------------------------------------------------------ the in-process transport: the ranks are threads of one process
 * (tests: the protocol at world sizes > 1 without a second GPU).  One shared mailbox; collectives are barriers over it. */
typedef struct {
	pthread_mutex_t mu; pthread_cond_t cv;
	int world, refs;
	/* collective slot */
	int64_t *coll; int coll_n, coll_arrived, coll_left; uint64_t coll_gen;
	/* point to point: one slot per (src -> dst 0) */
	struct { const void *buf; size_t n; int full; } *p2p;
} local_shared_t;
typedef struct { local_shared_t *S; int rank; } local_ctx_t;

/* every rank deposits n values, the last one releases all; *all gets world * n values (op 0) or the sums into mine (op 1) */
static int local_collective(local_ctx_t *c, int64_t *mine, int n, int64_t *all, int op)
{
	local_shared_t *S = c->S;
	uint64_t gen;
	int k, r;
	pthread_mutex_lock(&S->mu);
	while (S->coll_left > 0) pthread_cond_wait(&S->cv, &S->mu);   /* the previous collective is still being read */
	if (S->coll_arrived == 0) { S->coll = (int64_t*)realloc(S->coll, sizeof(int64_t) * (size_t)n * (size_t)S->world); S->coll_n = n; }
	if (S->coll_n != n) { pthread_mutex_unlock(&S->mu); return BSX_E_ARG; }
	memcpy(S->coll + (size_t)c->rank * (size_t)n, mine, sizeof(int64_t) * (size_t)n);
	gen = S->coll_gen;
	if (++S->coll_arrived == S->world) { S->coll_arrived = 0; S->coll_left = S->world; ++S->coll_gen; pthread_cond_broadcast(&S->cv); }
	else while (S->coll_gen == gen) pthread_cond_wait(&S->cv, &S->mu);
	if (op == 0) memcpy(all, S->coll, sizeof(int64_t) * (size_t)n * (size_t)S->world);
	else for (k = 0; k < n; ++k) { int64_t s = 0; for (r = 0; r < S->world; ++r) s += S->coll[(size_t)r * (size_t)n + k]; mine[k] = s; }
	if (--S->coll_left == 0) pthread_cond_broadcast(&S->cv);
	pthread_mutex_unlock(&S->mu);
	return BSX_OK;
}
static int local_all_gather(void *ctx, const int64_t *mine, int n, int64_t *all) { return local_collective((local_ctx_t*)ctx, (int64_t*)mine, n, all, 0); }
static int local_all_reduce(void *ctx, int64_t *buf, int n) { return local_collective((local_ctx_t*)ctx, buf, n, 0, 1); }
static int local_send(void *ctx, int dst, const void *buf, size_t n)
{
	local_ctx_t *c = (local_ctx_t*)ctx; local_shared_t *S = c->S;
	if (dst != 0) return BSX_E_ARG;
	pthread_mutex_lock(&S->mu);
	while (S->p2p[c->rank].full) pthread_cond_wait(&S->cv, &S->mu);
	S->p2p[c->rank].buf = buf; S->p2p[c->rank].n = n; S->p2p[c->rank].full = 1;
	pthread_cond_broadcast(&S->cv);
	while (S->p2p[c->rank].full) pthread_cond_wait(&S->cv, &S->mu);   /* the receiver has copied it */
	pthread_mutex_unlock(&S->mu);
	return BSX_OK;
}
static int local_recv_many(void *ctx, int n_src, const int *src, void *const *buf, const size_t *n)
{
	local_ctx_t *c = (local_ctx_t*)ctx; local_shared_t *S = c->S;
	int k, rc = BSX_OK;
	pthread_mutex_lock(&S->mu);
	for (k = 0; k < n_src; ++k) {
		while (!S->p2p[src[k]].full) pthread_cond_wait(&S->cv, &S->mu);
		if (S->p2p[src[k]].n != n[k]) rc = BSX_E_INTERNAL;   /* the size the sender announced is the size it sends */
		else memcpy(buf[k], S->p2p[src[k]].buf, n[k]);
		S->p2p[src[k]].full = 0;
		pthread_cond_broadcast(&S->cv);
	}
	pthread_mutex_unlock(&S->mu);
	return rc;
}
static void local_close(void *ctx)
{
	local_ctx_t *c = (local_ctx_t*)ctx; local_shared_t *S = c->S;
	int last;
	pthread_mutex_lock(&S->mu); last = --S->refs == 0; pthread_mutex_unlock(&S->mu);
	if (last) { pthread_mutex_destroy(&S->mu); pthread_cond_destroy(&S->cv); free(S->coll); free(S->p2p); free(S); }
	free(c);
}
/* out: `world` transports, one per thread-rank; each is closed by its own user */
BSX_API int bsx_transport_local(int world, bsx_transport_t *out)
{
	local_shared_t *S;
	int r;
	if (world < 1 || !out) return BSX_E_ARG;
	S = (local_shared_t*)calloc(1, sizeof(*S));
	pthread_mutex_init(&S->mu, 0); pthread_cond_init(&S->cv, 0);
	S->world = S->refs = world;
	S->p2p = calloc((size_t)world, sizeof(*S->p2p));
	for (r = 0; r < world; ++r) {
		local_ctx_t *c = (local_ctx_t*)calloc(1, sizeof(*c));
		c->S = S; c->rank = r;
		memset(&out[r], 0, sizeof(out[r]));
		out[r].ctx = c; out[r].rank = r; out[r].world = world;
		out[r].all_gather = local_all_gather; out[r].send = local_send; out[r].recv_many = local_recv_many; out[r].all_reduce_sum = local_all_reduce; out[r].close = local_close;
	}
	return BSX_OK;
}

/* ------------------------------------------------------------------ a socket transport: the ranks are processes of one node talking through
 * rank 0 over Unix-domain sockets (a star).  What the CPU suite runs the native multi-process path under (the CPU checker has no GPU to put
 * RCCL on), and a fallback on a box whose librccl cannot be loaded ("gather_transport=socket").  Every operation is a fixed exchange with
 * rank 0 in protocol order, so one stream socket per (rank, channel) carries it without framing. */
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
typedef struct { int rank, world, *fd; /* rank 0: fd[r] = the socket of rank r; others: fd[0] */ int lfd; char *path; int chan; } sock_ctx_t;
static int io_all(int fd, void *buf, size_t n, int wr)
{
	size_t done = 0;
	while (done < n) {
		ssize_t k = wr ? write(fd, (char*)buf + done, n - done) : read(fd, (char*)buf + done, n - done);
		if (k < 0) { if (errno == EINTR) continue; return BSX_E_IO; }
		if (k == 0) return BSX_E_IO;
		done += (size_t)k;
	}
	return BSX_OK;
}
static int sock_collective(sock_ctx_t *c, int64_t *mine, int n, int64_t *all, int op)
{
	const size_t nb = sizeof(int64_t) * (size_t)n;
	int r, k, rc;
	if (c->rank == 0) {
		int64_t *buf = (int64_t*)malloc(nb * (size_t)c->world);
		memcpy(buf, mine, nb);
		for (r = 1; r < c->world; ++r) if ((rc = io_all(c->fd[r], buf + (size_t)r * (size_t)n, nb, 0)) != BSX_OK) { free(buf); return rc; }
		if (op == 1) { for (k = 0; k < n; ++k) { int64_t s = 0; for (r = 0; r < c->world; ++r) s += buf[(size_t)r * (size_t)n + k]; mine[k] = s; } }
		for (r = 1; r < c->world; ++r) if ((rc = io_all(c->fd[r], op ? (void*)mine : (void*)buf, op ? nb : nb * (size_t)c->world, 1)) != BSX_OK) { free(buf); return rc; }
		if (op == 0) memcpy(all, buf, nb * (size_t)c->world);
		free(buf);
		return BSX_OK;
	}
	if ((rc = io_all(c->fd[0], mine, nb, 1)) != BSX_OK) return rc;
	return io_all(c->fd[0], op ? (void*)mine : (void*)all, op ? nb : nb * (size_t)c->world, 0);
}
static int sock_all_gather(void *ctx, const int64_t *mine, int n, int64_t *all) { return sock_collective((sock_ctx_t*)ctx, (int64_t*)mine, n, all, 0); }
static int sock_all_reduce(void *ctx, int64_t *buf, int n) { return sock_collective((sock_ctx_t*)ctx, buf, n, 0, 1); }
static int sock_send(void *ctx, int dst, const void *buf, size_t n) { sock_ctx_t *c = (sock_ctx_t*)ctx; return dst == 0 && c->rank != 0 ? io_all(c->fd[0], (void*)buf, n, 1) : BSX_E_ARG; }
static int sock_recv_many(void *ctx, int n_src, const int *src, void *const *buf, const size_t *n)
{
	sock_ctx_t *c = (sock_ctx_t*)ctx;
	int k, rc;
	for (k = 0; k < n_src; ++k) if ((rc = io_all(c->fd[src[k]], buf[k], n[k], 0)) != BSX_OK) return rc;
	return BSX_OK;
}
static void sock_close(void *ctx)
{
	sock_ctx_t *c = (sock_ctx_t*)ctx;
	int r;
	if (!c) return;
	for (r = 0; r < c->world; ++r) if (c->fd[r] >= 0) close(c->fd[r]);
	if (c->lfd >= 0) { close(c->lfd); if (c->path) unlink(c->path); }
	free(c->fd); free(c->path); free(c);
}
static int sock_make(int rank, int world, const char *path, int chan, bsx_transport_t *out)
{
	sock_ctx_t *c = (sock_ctx_t*)calloc(1, sizeof(*c));
	struct sockaddr_un a;
	int r, tries;
	c->rank = rank; c->world = world; c->lfd = -1; c->chan = chan;
	c->fd = (int*)malloc(sizeof(int) * (size_t)world);
	for (r = 0; r < world; ++r) c->fd[r] = -1;
	memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX;
	if (snprintf(a.sun_path, sizeof(a.sun_path), "%s.%d", path, chan) >= (int)sizeof(a.sun_path)) { sock_close(c); return BSX_E_ARG; }
	if (rank == 0) {
		c->path = strdup(a.sun_path);
		unlink(a.sun_path);
		c->lfd = socket(AF_UNIX, SOCK_STREAM, 0);
		if (c->lfd < 0 || bind(c->lfd, (struct sockaddr*)&a, sizeof(a)) != 0 || listen(c->lfd, world) != 0) { fprintf(stderr, "[bsx-gather] socket %s: %s\n", a.sun_path, strerror(errno)); sock_close(c); return BSX_E_IO; }
		for (r = 1; r < world; ++r) {
			int32_t who = -1;
			int fd = accept(c->lfd, 0, 0);
			if (fd < 0 || io_all(fd, &who, 4, 0) != BSX_OK || who < 1 || who >= world || c->fd[who] >= 0) { if (fd >= 0) close(fd); sock_close(c); return BSX_E_IO; }
			c->fd[who] = fd;
		}
	} else {
		int32_t who = rank;
		for (tries = 0; ; ++tries) { /* rank 0 may still be loading its index: ten minutes */
			struct timespec ts = {0, 100000000};
			c->fd[0] = socket(AF_UNIX, SOCK_STREAM, 0);
			if (c->fd[0] >= 0 && connect(c->fd[0], (struct sockaddr*)&a, sizeof(a)) == 0) break;
			if (c->fd[0] >= 0) { close(c->fd[0]); c->fd[0] = -1; }
			if (tries >= 6000) { fprintf(stderr, "[bsx-gather] rank %d: nobody listens on %s\n", rank, a.sun_path); sock_close(c); return BSX_E_IO; }
			nanosleep(&ts, 0);
		}
		if (io_all(c->fd[0], &who, 4, 1) != BSX_OK) { sock_close(c); return BSX_E_IO; }
	}
	memset(out, 0, sizeof(*out));
	out->ctx = c; out->rank = rank; out->world = world;
	out->all_gather = sock_all_gather; out->send = sock_send; out->recv_many = sock_recv_many; out->all_reduce_sum = sock_all_reduce; out->close = sock_close;
	return BSX_OK;
}
/* path: a name in a directory all ranks see (two sockets are made: <path>.0 for the gather, <path>.1 for the reductions) */
BSX_API int bsx_transport_socket(int rank, int world, const char *path, bsx_transport_t *gather, bsx_transport_t *reduce)
{
	int rc;
	if (world < 1 || rank < 0 || rank >= world || !path || (!gather && !reduce)) return BSX_E_ARG;
	if (gather && (rc = sock_make(rank, world, path, 0, gather)) != BSX_OK) return rc;
	if (reduce && (rc = sock_make(rank, world, path, 1, reduce)) != BSX_OK) { if (gather) { gather->close(gather->ctx); gather->ctx = 0; } return rc; }
	return BSX_OK;
}
