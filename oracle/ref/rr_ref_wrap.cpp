// TEST INFRASTRUCTURE.  Thin C entry point around the REFERENCE's own edit-distance code, compiled from the sources
// where they lie (/root/reference/scripts/read_recruitment/edlib/src/edlib.cpp, vendored edlib) into oracle/_ref/ by
// oracle/ref/Makefile — build container only; no reference source is copied into this repository.
// The call is exactly the one of scripts/read_recruitment/rr.cpp:74-79:
//   edlibAlign(unit, unit_len, read, read_len, edlibNewAlignConfig(threshold, EDLIB_MODE_HW, EDLIB_TASK_DISTANCE, NULL, 0))
#include <cstddef>

#include "edlib.h"

extern "C" int rr_ref_distance(const char* unit, int unit_len, const char* read, int read_len, int threshold) {
    EdlibAlignResult r = edlibAlign(unit, unit_len, read, read_len,
                                    edlibNewAlignConfig(threshold, EDLIB_MODE_HW, EDLIB_TASK_DISTANCE, NULL, 0));
    const int d = r.status == EDLIB_STATUS_OK ? r.editDistance : -2;
    edlibFreeAlignResult(r);
    return d;
}
