// gather_rccl.hip -- the gather's communication layer over RCCL (csrc/host/gather.c, include/bsx.h: bsx_transport_t).
//
// One process per GPU; the chunk records move GPU to GPU over xGMI: a rank's chunk goes pinned host -> its device staging buffer -> ncclSend,
// rank 0 posts the round's ncclRecv calls as ONE group (xGMI is point to point: every sender has its own link into rank 0, so the round's
// transfers run side by side), each into the sender's own staging buffer, then device -> pinned host -> the caller's buffer.  Sizes travel as
// an ncclAllGather of int64, the insert-size histograms of ranks sharing a chunk as an ncclAllReduce on a communicator of its own (it is called
// from the aligner's thread while the rounds run on another).
// librccl is loaded at run time (dlopen): the library itself does not depend on it, and a box without RCCL gets BSX_E_NODEVICE here and
// nowhere else.  The unique ids reach the other ranks through a file next to the output (rank 0 writes it under a temporary name and renames
// it; the others wait for it to appear).
// Not exercised between two GPUs by the builder (one-GPU boxes only): the world-size-1 paths (initialisation, all-gather, all-reduce, the
// id file) run in the -m gpu suite, the protocol above it at world sizes 2-4 over the in-process transport in the CPU suite.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <vector>
#include "bsx.h"
#define BSX_API __attribute__((visibility("default")))

#include <rccl/rccl.h>   // types and enumerators only: every function is looked up in the library loaded at run time (no link-time dependency)
struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)(void) = nullptr;
	ncclResult_t (*GroupEnd)(void) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load(void)
{
	if (g_rccl.lib) return BSX_OK;
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!h) { fprintf(stderr, "[bsx-rccl] librccl cannot be loaded: %s\n", dlerror()); return BSX_E_NODEVICE; }
#define SYM(f) do { *(void**)&g_rccl.f = dlsym(h, "nccl" #f); if (!g_rccl.f) { fprintf(stderr, "[bsx-rccl] librccl has no nccl" #f "\n"); dlclose(h); return BSX_E_NODEVICE; } } while (0)
	SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(AllGather); SYM(AllReduce); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
	g_rccl.lib = h;
	return BSX_OK;
}
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "[bsx-rccl] %s failed: %s (%s:%d)\n", #x, g_rccl.GetErrorString(r_), __FILE__, __LINE__); return BSX_E_NODEVICE; } } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[bsx-rccl] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return BSX_E_NODEVICE; } } while (0)

struct Stage { void *dev = nullptr, *pin = nullptr; size_t cap = 0; };
struct RcclCtx {
	int rank = 0, world = 1, device = 0;
	ncclComm_t comm = nullptr;
	hipStream_t st = nullptr;
	Stage own;                 // this rank's outgoing chunk / the collectives' buffers
	std::vector<Stage> from;   // rank 0: one staging pair per sender
};
static int stage_reserve(Stage &s, size_t n)
{
	if (s.cap >= n) return BSX_OK;
	if (s.dev) (void)hipFree(s.dev);
	if (s.pin) (void)hipHostFree(s.pin);
	s.dev = s.pin = nullptr; s.cap = 0;
	const size_t cap = n + (n >> 2) + ((size_t)1 << 20);
	HCHK(hipMalloc(&s.dev, cap));
	HCHK(hipHostMalloc(&s.pin, cap, hipHostMallocDefault));
	s.cap = cap;
	return BSX_OK;
}

static int r_all_gather(void *c_, const int64_t *mine, int n, int64_t *all)
{
	RcclCtx *c = (RcclCtx*)c_;
	HCHK(hipSetDevice(c->device));
	const size_t nb = (size_t)n * 8;
	int rc;
	if ((rc = stage_reserve(c->own, nb * (size_t)(c->world + 1))) != BSX_OK) return rc;
	memcpy(c->own.pin, mine, nb);
	HCHK(hipMemcpyAsync(c->own.dev, c->own.pin, nb, hipMemcpyHostToDevice, c->st));
	NCHK(g_rccl.AllGather(c->own.dev, (char*)c->own.dev + nb, (size_t)n, ncclInt64, c->comm, c->st));
	HCHK(hipMemcpyAsync((char*)c->own.pin + nb, (char*)c->own.dev + nb, nb * (size_t)c->world, hipMemcpyDeviceToHost, c->st));
	HCHK(hipStreamSynchronize(c->st));
	memcpy(all, (char*)c->own.pin + nb, nb * (size_t)c->world);
	return BSX_OK;
}
static int r_all_reduce(void *c_, int64_t *buf, int n)
{
	RcclCtx *c = (RcclCtx*)c_;
	HCHK(hipSetDevice(c->device));
	const size_t nb = (size_t)n * 8;
	int rc;
	if ((rc = stage_reserve(c->own, nb)) != BSX_OK) return rc;
	memcpy(c->own.pin, buf, nb);
	HCHK(hipMemcpyAsync(c->own.dev, c->own.pin, nb, hipMemcpyHostToDevice, c->st));
	NCHK(g_rccl.AllReduce(c->own.dev, c->own.dev, (size_t)n, ncclInt64, ncclSum, c->comm, c->st));
	HCHK(hipMemcpyAsync(c->own.pin, c->own.dev, nb, hipMemcpyDeviceToHost, c->st));
	HCHK(hipStreamSynchronize(c->st));
	memcpy(buf, c->own.pin, nb);
	return BSX_OK;
}
static int r_send(void *c_, int dst, const void *buf, size_t n)
{
	RcclCtx *c = (RcclCtx*)c_;
	HCHK(hipSetDevice(c->device));
	int rc;
	if ((rc = stage_reserve(c->own, n)) != BSX_OK) return rc;
	memcpy(c->own.pin, buf, n);   // (pageable -> pinned here: handing pageable memory to the copy makes the runtime pin it page by page)
	HCHK(hipMemcpyAsync(c->own.dev, c->own.pin, n, hipMemcpyHostToDevice, c->st));
	NCHK(g_rccl.Send(c->own.dev, n, ncclChar, dst, c->comm, c->st));
	HCHK(hipStreamSynchronize(c->st));
	return BSX_OK;
}
static int r_recv_many(void *c_, int n_src, const int *src, void *const *buf, const size_t *n)
{
	RcclCtx *c = (RcclCtx*)c_;
	HCHK(hipSetDevice(c->device));
	int rc;
	if (c->from.size() < (size_t)c->world) c->from.resize((size_t)c->world);
	for (int k = 0; k < n_src; ++k) if ((rc = stage_reserve(c->from[(size_t)src[k]], n[k])) != BSX_OK) return rc;
	NCHK(g_rccl.GroupStart());
	for (int k = 0; k < n_src; ++k) NCHK(g_rccl.Recv(c->from[(size_t)src[k]].dev, n[k], ncclChar, src[k], c->comm, c->st));
	NCHK(g_rccl.GroupEnd());
	for (int k = 0; k < n_src; ++k) HCHK(hipMemcpyAsync(c->from[(size_t)src[k]].pin, c->from[(size_t)src[k]].dev, n[k], hipMemcpyDeviceToHost, c->st));
	HCHK(hipStreamSynchronize(c->st));
	for (int k = 0; k < n_src; ++k) memcpy(buf[k], c->from[(size_t)src[k]].pin, n[k]);
	return BSX_OK;
}
static void stage_free(Stage &s) { if (s.dev) (void)hipFree(s.dev); if (s.pin) (void)hipHostFree(s.pin); s = Stage(); }
static void r_close(void *c_)
{
	RcclCtx *c = (RcclCtx*)c_;
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->st) (void)hipStreamSynchronize(c->st);
	if (c->comm) (void)g_rccl.CommDestroy(c->comm);
	stage_free(c->own);
	for (auto &s : c->from) stage_free(s);
	if (c->st) (void)hipStreamDestroy(c->st);
	delete c;
}

// the unique ids through a file: rank 0 writes all of them under a temporary name and renames it; the others wait for the file
static int ids_exchange(int rank, int world, const char *path, ncclUniqueId *ids, int n_ids)
{
	const size_t nb = sizeof(ncclUniqueId) * (size_t)n_ids;
	if (world == 1) { for (int k = 0; k < n_ids; ++k) NCHK(g_rccl.GetUniqueId(&ids[k])); return BSX_OK; }
	if (!path || !*path) { fprintf(stderr, "[bsx-rccl] %d ranks need a path for the unique ids\n", world); return BSX_E_ARG; }
	if (rank == 0) {
		for (int k = 0; k < n_ids; ++k) NCHK(g_rccl.GetUniqueId(&ids[k]));
		std::vector<char> tmp(strlen(path) + 16);
		snprintf(tmp.data(), tmp.size(), "%s.tmp%d", path, (int)getpid());
		FILE *f = fopen(tmp.data(), "wb");
		if (!f || fwrite(ids, 1, nb, f) != nb || fclose(f) != 0 || rename(tmp.data(), path) != 0) { fprintf(stderr, "[bsx-rccl] writing %s failed: %s\n", path, strerror(errno)); return BSX_E_IO; }
		return BSX_OK;
	}
	for (int tries = 0; tries < 6000; ++tries) { // ten minutes: rank 0 may still be loading its index
		FILE *f = fopen(path, "rb");
		if (f) {
			const size_t got = fread(ids, 1, nb, f);
			fclose(f);
			if (got == nb) return BSX_OK;
		}
		struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr);
	}
	fprintf(stderr, "[bsx-rccl] rank %d: %s (the unique ids of rank 0) did not appear\n", rank, path);
	return BSX_E_IO;
}

static int make_comm(int rank, int world, int device, const ncclUniqueId &id, bsx_transport_t *out)
{
	RcclCtx *c = new RcclCtx();
	c->rank = rank; c->world = world; c->device = device;
	if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { delete c; return BSX_E_NODEVICE; }
	ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
	if (r != ncclSuccess) { fprintf(stderr, "[bsx-rccl] ncclCommInitRank (rank %d of %d) failed: %s\n", rank, world, g_rccl.GetErrorString(r)); (void)hipStreamDestroy(c->st); delete c; return BSX_E_NODEVICE; }
	memset(out, 0, sizeof(*out));
	out->ctx = c; out->rank = rank; out->world = world;
	out->all_gather = r_all_gather; out->send = r_send; out->recv_many = r_recv_many; out->all_reduce_sum = r_all_reduce; out->close = r_close;
	return BSX_OK;
}

extern "C" BSX_API int bsx_transport_rccl(int rank, int world, int device, const char *id_path, bsx_transport_t *gather, bsx_transport_t *reduce)
{
	if (world < 1 || rank < 0 || rank >= world || (!gather && !reduce)) return BSX_E_ARG;
	int n_dev = 0, rc;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { fprintf(stderr, "[bsx-rccl] no HIP device %d\n", device); return BSX_E_NODEVICE; }
	if ((rc = rccl_load()) != BSX_OK) return rc;
	HCHK(hipSetDevice(device));
	ncclUniqueId ids[2];
	if ((rc = ids_exchange(rank, world, id_path, ids, 2)) != BSX_OK) return rc;
	if (gather && (rc = make_comm(rank, world, device, ids[0], gather)) != BSX_OK) return rc;
	if (reduce && (rc = make_comm(rank, world, device, ids[1], reduce)) != BSX_OK) { if (gather) { gather->close(gather->ctx); gather->ctx = nullptr; } return rc; }
	// every rank has read the ids once its communicators are up (ncclCommInitRank returns when all ranks have joined): rank 0 removes the file
	if (world > 1 && rank == 0) (void)unlink(id_path);
	return BSX_OK;
}
