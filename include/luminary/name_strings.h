/* Forwarding header: a frontend written against the reference's <luminary/name_strings.h> (reference include/luminary/name_strings.h) compiles against
 * libluminary_amd.so unchanged. Every declaration of the reference's public headers lives in ../luminary_amd.h. */
#ifndef LUMINARY_AMD_FORWARD_NAME_STRINGS_H
#define LUMINARY_AMD_FORWARD_NAME_STRINGS_H
#include "../luminary_amd.h"
#endif
