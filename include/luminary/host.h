/* Forwarding header: a frontend written against the reference's <luminary/host.h> (reference include/luminary/host.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_HOST_H
#define LUMINARY_AMD_FORWARD_HOST_H
#include "../luminary_amd.h"
#endif
