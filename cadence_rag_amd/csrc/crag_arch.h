// crag_arch.h -- included by every .hip of the library.
// gfx950 only: the kernels use up to 154 KiB of LDS per workgroup (the other CDNA parts have 64 KiB), MFMA shapes and
// v_permlane swaps that exist on CDNA4 alone.  `make ARCH=...` with anything else stops here instead of building a
// library whose launches fail at run time.
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libcrag_dense is written for gfx950 (MI355X) only"
#endif
