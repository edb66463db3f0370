/* mi_common.h -- shared conventions of the MI355X (gfx950) hot-path library libmi_ipp.so.
 *
 * C ABI, plain pointers and sizes.  Unless a parameter is marked [host], every data pointer is a
 * DEVICE pointer on HIP device `dev` (e.g. a torch-ROCm tensor's data_ptr()).  `stream` is a
 * hipStream_t passed as void* (NULL = the device's default stream).  Calls only enqueue work on
 * `stream` unless their comment says "synchronises".  Every function returns MI_OK (0) or a
 * negative mi_status and never calls exit()/abort(); the message of the last failure on the calling
 * thread is returned by mi_last_error().  The library is re-entrant per (device, stream): it keeps
 * no static device buffers bound to a call (contrast compute_funcs.cu:621-629 of the reference); FFT
 * plans are cached per device under a mutex; scratch memory that a call releases is kept in a per-device
 * pool and reused by the next request of the same size (see mi_release_cached_memory).
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
    MI_ERR_FFT = -3,         /* a rocFFT call failed, or a rocFFT plan did not transform (every engine on that route checks itself once) */
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
/* The library keeps the device memory its calls release (>= 1 MiB blocks) and reuses it for later requests of the same size:
 * a per-block pipeline (edge taper, RL context, ...) otherwise pays hipMalloc/hipFree of tens of GB per stage.  Cached blocks are
 * given back automatically when an allocation fails; mi_release_cached_memory returns them to the driver now (dev < 0: all
 * devices) and reports the bytes released; MI_NO_MEMORY_POOL=1 in the environment disables the pool. */
size_t mi_release_cached_memory(int dev);
size_t mi_cached_memory_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_COMMON_H */
