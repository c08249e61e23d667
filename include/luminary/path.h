/* Forwarding header: a frontend written against the reference's <luminary/path.h> (reference include/luminary/path.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_PATH_H
#define LUMINARY_AMD_FORWARD_PATH_H
#include "../luminary_amd.h"
#endif
