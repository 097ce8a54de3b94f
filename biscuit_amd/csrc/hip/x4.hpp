// x4.hpp -- shared by k_ext4.hip (extensions four to a wavefront) and k_extl.hip (a lane per extension): the job record k_x4prep writes
// and the reference bases of an extension's rows as 2-bit fields in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_common.hpp"

struct X4Job {                    // 48 bytes
	long long s_rbeg, rmax0, rmax1;   // the seed's reference start; the chain's window (memchain.c:585-610)
	unsigned long long ext_at;        // byte offset of the chain's RgXExt in the export pool
	unsigned int qoff;                // the read in the chunk's read buffer
	short l_query, s_qbeg, s_len; unsigned char parent, pad;
	int si;                           // the seed's index in its list
};

#define X4_NARROW 32     // seeds shorter than this are queued for k_extl (chance matches of a 3-letter 19-mer: short, narrow extensions)

// 2-bit fields of a word in reverse order
__device__ __forceinline__ uint32_t x4_rev2(uint32_t x) { x = __builtin_bitreverse32(x); return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); }

// The reference bases of rows r0 .. r0 + 47 of an extension (fewer when the target ends: need = rows left, >= 1) as 2-bit fields, row
// r0 + m at bits 2m of the 96-bit number y2:y1:y0.  F = forward-strand coordinate of row r0's base, fd = its step per row; the rows
// lie on one strand, so they are consecutive fields of pac read up or down (bns_get_seq, bntseq.c:402-422).  Four aligned dwords of
// pac (64 bases) hold any 48 consecutive ones; pac is padded past its end.
__device__ __forceinline__ void x4_bases(const uint8_t *pac, long long F, int fd, int need, uint32_t &y0, uint32_t &y1, uint32_t &y2)
{
	(void)need;   // (the rows that exist lie inside the window, i.e. at coordinates >= 0; the fields of the others are never read)
	const long long P0 = (fd > 0 ? F : (F - 47 > 0 ? F - 47 : 0)) & ~15ll;   // first base of the first dword
	const uint4 v = *reinterpret_cast<const uint4*>(pac + (P0 >> 2));
	// base q of pac sits at bits 126 - 2 (q - P0) of the bytes read as one big-endian number
	const unsigned __int128 B = (unsigned __int128)__builtin_bswap32(v.x) << 96 | (unsigned __int128)__builtin_bswap32(v.y) << 64 |
	                            (unsigned __int128)__builtin_bswap32(v.z) << 32 | (unsigned __int128)__builtin_bswap32(v.w);
	if (fd > 0) { // row m = base F + m: the top 96 bits after the shift, field order reversed
		const unsigned __int128 S = B << (2 * (int)(F - P0));
		y0 = x4_rev2((uint32_t)(S >> 96)); y1 = x4_rev2((uint32_t)(S >> 64)); y2 = x4_rev2((uint32_t)(S >> 32));
	} else {      // row m = base F - m: base F to bits 0
		const unsigned __int128 S = B >> (126 - 2 * (int)(F - P0));
		y0 = (uint32_t)S; y1 = (uint32_t)(S >> 32); y2 = (uint32_t)(S >> 64);
	}
}

