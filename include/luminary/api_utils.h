/* Forwarding header: a frontend written against the reference's <luminary/api_utils.h> (reference include/luminary/api_utils.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_API_UTILS_H
#define LUMINARY_AMD_FORWARD_API_UTILS_H
#include "../luminary_amd.h"
#endif
