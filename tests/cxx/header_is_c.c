/* include/rccl_amd.h must be usable from plain C (gcc -std=c11 -fsyntax-only -Wall -Werror). */
#include "rccl_amd.h"
int uses_the_api(void) {
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  ncclUniqueId id;
  int v = 0;
  (void)cfg; (void)id;
  return (int)ncclGetVersion(&v) + (int)sizeof(ncclComm_t) + (int)ncclMaxRedOp;
}
