// Sources compiled in PARTS. The first launch of any kernel of a code object loads the WHOLE object (~3 ms per MB on the MI355X box,
// rocprofv3 --hip-trace of tools/first_call.py: 5.0 ms for the first K-means launch out of a 1.5 MB object); a fit uses one
// dimension and one component count, so the template-heavy files are compiled several times with -DMLHIP_PART=n (ml_amd/csrc/Makefile),
// every part holding the instantiations of a few shapes and exporting  <entry>_part<n>(...)  -- the dispatcher (in part 1) picks the
// part by shape. One code object per part: the first fit of a process loads a few hundred KB instead of several MB.
#pragma once
#ifndef MLHIP_PART
#error "this file is compiled in parts: -DMLHIP_PART=n (see ml_amd/csrc/Makefile)"
#endif
#define MLHIP_CAT2(a, b) a##b
#define MLHIP_CAT(a, b) MLHIP_CAT2(a, b)
#define MLHIP_PART_FN(name) MLHIP_CAT(name##_part, MLHIP_PART)
