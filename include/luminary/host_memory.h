/* Forwarding header: a frontend written against the reference's <luminary/host_memory.h> (reference include/luminary/host_memory.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_HOST_MEMORY_H
#define LUMINARY_AMD_FORWARD_HOST_MEMORY_H
#include "../luminary_amd.h"
#endif
