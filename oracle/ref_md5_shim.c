/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Compiles the reference's own src/Md5.c where it lies (-I/root/reference/src) into
 * oracle/_ref/libptref.so so that the host mirror's RFC 1321 implementation (pathtracer_amd/host/Md5.hpp, which names the hierarchy
 * cache file like RayTracer::computeMD5, reference src/RayTracer.cpp:57-66) is checked against the reference's function, not only
 * against the RFC's published vectors.
 * Data model: Md5.c declares "typedef unsigned long int UINT4" as its 32-bit type (src/Md5.c:42-43).  That holds on the reference's
 * platform (MSVC, LLP64); on LP64 Linux `long` is 64 bits and the file, compiled unchanged, returns digests that are not MD5
 * (MD5("abc") = 7999dc75... instead of 90015098...).  So the file is compiled here with `long` defined away inside it -- the type
 * becomes `unsigned int`, 32 bits, as on its own platform -- after <stddef.h> (its only include) has been read with the real keyword. */
#include <stddef.h>
#define long
#include "Md5.c"
#undef long
