// Per-street fused engine (prl_st.h): the street kernels of spec 33 (prl_fhp.h: PrlFhpSpec33), one translation unit per spec so
// that the specs compile in parallel. The kernels are in prl_st_pass.inc.
#include <string>
#include <type_traits>

#include "prl_device.h"
#include "prl_kernels.h"
#include "prl_st.h"
#include "prl_st_specs.h"

#define ST_SPEC PrlFhpSpec33
namespace st_spec33 {
#include "prl_st_pass.inc"
}
