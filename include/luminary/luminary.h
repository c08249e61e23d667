/* Forwarding header: a frontend written against the reference's <luminary/luminary.h> (reference include/luminary/luminary.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_LUMINARY_H
#define LUMINARY_AMD_FORWARD_LUMINARY_H
#include "../luminary_amd.h"
#endif
