/* Forwarding header: a frontend written against the reference's <luminary/structs.h> (reference include/luminary/structs.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_STRUCTS_H
#define LUMINARY_AMD_FORWARD_STRUCTS_H
#include "../luminary_amd.h"
#endif
