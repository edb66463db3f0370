/* mi_common.h -- shared conventions of the MI355X (gfx950) hot-path library libmi_ipp.so.
 *
 * C ABI, plain pointers and sizes.  Unless a parameter is marked [host], every data pointer is a
 * DEVICE pointer on HIP device `dev` (e.g. a torch-ROCm tensor's data_ptr()).  `stream` is a
 * hipStream_t passed as void* (NULL = the device's default stream).  Calls only enqueue work on
 * `stream` unless their comment says "synchronises".  Every function returns MI_OK (0) or a
 * negative mi_status and never calls exit()/abort(); the message of the last failure on the calling
 * thread is returned by mi_last_error().  The library is re-entrant per (device, stream): it keeps
 * no static device buffers (contrast compute_funcs.cu:621-629 of the reference); FFT plans are
 * cached per device under a mutex.
 *
 * Array convention: volumes are C-order (Z, Y, X) with X fastest == MATLAB [X,Y,Z] column-major
 * (conv3d_gpu.cu:93,98) == TeraStitcher (k, i, j) slice/row/column (CrossMIPs.h:109).  Dimensions
 * are passed X first (nx, ny, nz / kx, ky, kz) like the reference MEX files do.
 */
#ifndef MI_COMMON_H
#define MI_COMMON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MI_OK = 0,
    MI_ERR_INVALID = -1,     /* bad argument (the reference's mexErrMsgIdAndTxt / iom::exception cases) */
    MI_ERR_HIP = -2,         /* a HIP runtime call failed */
    MI_ERR_FFT = -3,         /* a rocFFT call failed */
    MI_ERR_NOMEM = -4,       /* workspace too small / allocation failed */
    MI_ERR_UNSUPPORTED = -5  /* valid in the reference but not built here (e.g. NCC `enhance`) */
} mi_status;

/* message of the last error raised on this thread ("" if none) */
const char* mi_last_error(void);
/* number of visible HIP devices, or a negative mi_status */
int mi_device_count(void);
/* library ABI version (bumped when a signature changes) */
int mi_abi_version(void);
/* blocks until `stream` has drained (hipStreamSynchronize) */
int mi_stream_synchronize(int dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_COMMON_H */
