/* Forwarding header: code written against the reference's installed header, #include <rccl/rccl.h>
 * (generated there from src/nccl.h.in), builds against this library with -I<repo>/include. */
#ifndef RCCL_AMD_FORWARD_RCCL_H_
#define RCCL_AMD_FORWARD_RCCL_H_
#include "../rccl_amd.h"
#endif
