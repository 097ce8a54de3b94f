// devbuf.hpp -- growable device / pinned host buffers shared by shim.hip and the index builder
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
extern "C" {
#include "bsx_core.h"
}

struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	int reserve(size_t n) {
		if (n <= cap) return BSX_OK;
		if (p) (void)hipFree(p);
		size_t want = n + (n >> 2) + 4096;
		if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; cap = 0; fprintf(stderr, "[bsx-hip] hipMalloc(%zu) failed\n", want); return BSX_E_NOMEM; }
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
		if (p) (void)hipHostFree(p);
		size_t want = n + (n >> 2) + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; cap = 0; return BSX_E_NOMEM; }
		cap = want;
		return BSX_OK;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
