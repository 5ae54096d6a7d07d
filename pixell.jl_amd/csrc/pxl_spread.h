// pxl_spread.h -- the placement probe; included by pxl_kernels.hip.
//
// Round 3 found what makes "the same" 7-22 GB destination 10-17 % faster in some places than in others
// (tools/research/exp_xcd_affinity.cpp, exp_vmm_vs_malloc.cpp; profiles/r03_xcd_classes.txt): the memory of a hipMalloc'ed
// allocation falls into three classes -- any two 1 GiB windows are cleanly either "of one class" or "of different classes", an
// equivalence relation with three classes of up to 96 GiB each, i.e. the thirds of the 288 GiB part (the three ranks of its
// 12-high HBM3E stacks is our reading; nothing here depends on the name) -- and a kernel that keeps several far-apart WRITE
// fronts going (the reprojection has eight, one per XCD) stores at 5.8-6.0 TB/s when all its fronts lie in one class and at
// 6.8-7.1 TB/s when they are split over two.  A single XCD writes everywhere at the same 1.30 TB/s, reads do not care, and a
// plain fill (one front) does not see it.  A map allocated with hipMalloc is made of large physically contiguous blocks and
// normally lies inside one class; only when a block boundary between two classes happened to fall into the destination did
// rounds 1 and 2 see the fast case ("destination above the source in one allocation" for the 45 GB pair).
//
// The probe below IS the definition of the classes: eight store fronts, four in window a, four in window b, one per XCD.  A host
// that wants its destination to straddle a class boundary can map an allocation with it (pixell.jl_amd/placement.py does: one
// probe per 2 GiB, 0.3 ms each).  It overwrites both windows with zeros.
#pragma once

// blocks with (blockIdx & 7) == v share an XCD and write piece (v >> 1) of window a (v even) or b (v odd)
__global__ __launch_bounds__(256) void k_spread_probe(char* a, char* b, size_t piece_bytes) {
    const int v = blockIdx.x & 7;
    const size_t j = blockIdx.x >> 3, nj = gridDim.x >> 3;
    uint4* p = reinterpret_cast<uint4*>(((v & 1) ? b : a) + (size_t)(v >> 1) * piece_bytes);
    const size_t n = piece_bytes / 16, chunk = 4096;                   // 64 KiB per block and trip: a moving front
    const uint4 val = make_uint4(0, 0, 0, 0);
    for (size_t q = j; q * chunk < n; q += nj)
        for (size_t i = q * chunk + threadIdx.x; i < (q + 1) * chunk && i < n; i += 256) p[i] = val;
}
