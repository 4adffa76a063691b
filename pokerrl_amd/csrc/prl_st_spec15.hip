// Per-street fused engine (prl_st.h): the street kernels of spec 15 (prl_fhp.h: PrlFhpSpec15), one translation unit per spec so
// that the specs compile in parallel. The kernels are in prl_st_pass.inc.
#include <string>
#include <type_traits>

#include "prl_device.h"
#include "prl_kernels.h"
#include "prl_st.h"
#include "prl_st_specs.h"

#define ST_SPEC PrlFhpSpec15
namespace st_spec15 {
#include "prl_st_pass.inc"
}
