// 4x4x8-tile instantiations of the LDS-halo weight-gradient kernel: the same source as wgrad_halo.hip, its own translation
// unit so that it can be compiled with the instruction scheduler that suits it (build.py: PER_FILE_FLAGS).
#define WH_T44_UNIT 1
#include "wgrad_halo.hip"
