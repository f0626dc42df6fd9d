/* A C caller built from include/ssasr.h alone: links libssasr_hip.so, checks the ABI version and
 * that argument errors come back negative without touching a GPU.  Compiled and run by
 * tests/test_host_cpu.py::test_a_c_caller_builds_against_the_header (gcc, no HIP headers). */
#include <stdio.h>
#include "ssasr.h"

int main(void) {
  int v = ssasr_abi_version();
  printf("abi %d\n", v);
  /* every size 0 / pointer NULL: argument error, nothing launched */
  int rc_wgrad = ssasr_bilstm_wgrad(NULL, NULL, 0, 0, NULL, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                                    NULL, NULL, /*accumulate*/ 1, /*stream*/ NULL);
  int rc_attn = ssasr_attn_step_fwd(NULL, NULL, NULL, NULL, NULL, 0, 0, 0, 0, 0, NULL, NULL, NULL,
                                    /*ws*/ NULL, /*ws_phase*/ 0, /*ws_status*/ NULL, /*stream*/ NULL);
  int rc_dec = ssasr_decoder_fwd(NULL, NULL);
  int opt = 12345;
  int rc_opt = ssasr_get_option("SSASR_NO_SUCH_SWITCH", &opt);
  printf("wgrad %d attn %d decoder %d option %d\n", rc_wgrad, rc_attn, rc_dec, rc_opt);
  return (v > 0 && rc_wgrad < 0 && rc_attn < 0 && rc_dec < 0 && rc_opt == -1) ? 0 : 1;
}
