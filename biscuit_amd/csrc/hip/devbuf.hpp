// devbuf.hpp -- growable device / pinned host buffers shared by shim.hip and the index builder
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" {
#include "bsx_core.h"
}

// hipFree / hipHostFree wait for the whole device: when a per-chunk buffer of one lane grows while the kernels of three other chunks are
// running, the call returns seconds later (measured: K5/K6 batches of 40 ms taking 1.2 s during the first chunks of every lane).  A
// buffer that grows therefore leaves its old block on a list that is emptied when the device is closed.
#include <mutex>
#include <vector>
#include <utility>
struct DevbufDeferred { std::mutex mu; std::vector<std::pair<int, void*>> dev, host; };   // (device ordinal current when the block was retired, block)
inline DevbufDeferred &devbuf_deferred() { static DevbufDeferred d; return d; }
inline int devbuf_current() { int o = 0; if (hipGetDevice(&o) != hipSuccess) { (void)hipGetLastError(); o = -1; } return o; }
// frees what the buffers of device `ordinal` left behind (the current device must be that one): another open device's lanes may be in
// the middle of a chunk, and hipFree waits for whatever runs on the device the block belongs to
inline void devbuf_drain(int ordinal) {
	DevbufDeferred &d = devbuf_deferred();
	std::lock_guard<std::mutex> g(d.mu);
	size_t k = 0;
	for (auto &q : d.dev) { if (q.first == ordinal || q.first < 0) (void)hipFree(q.second); else d.dev[k++] = q; }
	d.dev.resize(k);
	k = 0;
	for (auto &q : d.host) { if (q.first == ordinal || q.first < 0) (void)hipHostFree(q.second); else d.host[k++] = q; }
	d.host.resize(k);
}

struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t n) {
		if (n <= cap) return BSX_OK;
		if (p) { DevbufDeferred &d = devbuf_deferred(); std::lock_guard<std::mutex> g(d.mu); d.dev.push_back(std::make_pair(devbuf_current(), p)); p = nullptr; }
		// room to grow into: half again for the small buffers (their sizes follow the data), a sixteenth for those of a gigabyte and more (pools
		// sized from the chunk's read count: the four lanes of a stream held 60 GB of slack between them)
		size_t want = n + (n >= ((size_t)1 << 30) ? n >> 4 : n >> 1) + 4096;
		if (hipMalloc(&p, want) != hipSuccess) {
			size_t fr = 0, tot = 0; (void)hipGetLastError(); (void)hipMemGetInfo(&fr, &tot);
			p = nullptr; cap = 0; fprintf(stderr, "[bsx-hip] hipMalloc(%zu) failed (%zu of %zu bytes free)\n", want, fr, tot); return BSX_E_NOMEM;
		}
		{ static const bool tr = getenv("BSX_TRACE_ALLOC") != nullptr; if (tr && want >= ((size_t)1 << 30)) fprintf(stderr, "[bsx-hip] DevBuf %p: %.1f GB (was %.1f)\n", (void*)this, want / 1073741824.0, cap / 1073741824.0); }
		cap = want;
		return BSX_OK;
	}
	// exactly n bytes (the index builder's arrays are tens of GB: no slack)
	int reserve_exact(size_t n) {
		if (n <= cap) return BSX_OK;
		if (p) (void)hipFree(p);
		if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; cap = 0; (void)hipGetLastError(); fprintf(stderr, "[bsx-hip] hipMalloc(%zu) failed\n", n); return BSX_E_NOMEM; }
		cap = n;
		return BSX_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostBuf {   // pinned host staging (D2H/H2D at full PCIe rate, no hidden bounce copy)
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t n) {
		if (n <= cap) return BSX_OK;
		if (p) { DevbufDeferred &d = devbuf_deferred(); std::lock_guard<std::mutex> g(d.mu); d.host.push_back(std::make_pair(devbuf_current(), p)); p = nullptr; }
		size_t want = n + (n >> 1) + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; return BSX_E_NOMEM; }
		cap = want;
		return BSX_OK;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
