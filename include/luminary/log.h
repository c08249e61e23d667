/* Forwarding header: a frontend written against the reference's <luminary/log.h> (reference include/luminary/log.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_LOG_H
#define LUMINARY_AMD_FORWARD_LOG_H
#include "../luminary_amd.h"
#endif
