/* pthost_internal.h - shared helpers of libpthost (not part of the C-ABI). */
#ifndef PTHOST_INTERNAL_H
#define PTHOST_INTERNAL_H

#if defined(__GNUC__)
__attribute__((format(printf, 1, 2)))
#endif
void pthost_set_error(const char *fmt, ...);

#endif
