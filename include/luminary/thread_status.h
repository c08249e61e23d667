/* Forwarding header: a frontend written against the reference's <luminary/thread_status.h> (reference include/luminary/thread_status.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_THREAD_STATUS_H
#define LUMINARY_AMD_FORWARD_THREAD_STATUS_H
#include "../luminary_amd.h"
#endif
