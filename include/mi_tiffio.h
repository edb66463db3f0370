/* TIFF series I/O of the block pipeline (host code, no GPU): the slices decwrap.py reads its boxes from and writes its result to.
 *
 * Replaces, for the files the pipeline itself meets, LsDeconvolveMultiGPU/load_bl_tif.cpp (a box of a folder of 2-D TIFF slices,
 * one libtiff handle per thread) and save_bl_tif.cpp (one slice per task on all cores, Adobe deflate at ZIPQUALITY 1, predictor 1:
 * save_bl_tif.cpp:336-346, called with 'deflate' by LsDeconv.m:1140-1145).  Python's Pillow writes 29 MB/s of deflate TIFF and holds
 * the interpreter lock while it does: 17 GB of result would take ten minutes where the deconvolution takes five seconds.
 *
 * Scope of the reader ("fast" files): classic little-endian TIFF, one sample per pixel, 8 / 16 / 32 bits (unsigned integer or IEEE
 * float), strips (any RowsPerStrip), compression none / Adobe deflate (8) / deflate (32946), predictor none or horizontal
 * differencing (2).  Everything else -- tiles, LZW, BigTIFF, big-endian, palettes -- is reported as not fast and left to the caller's
 * general reader (brickio.py falls back to Pillow).  The writer produces exactly such files.
 *
 * All functions return MI_OK or an mi_status (mi_common.h); the message is available through mi_last_error(). */
#ifndef MI_TIFFIO_H
#define MI_TIFFIO_H

#include "mi_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Width, height and sample type (1 = uint8, 2 = uint16, 4 = float32; 0 = none of these) of the first image of a file, and
 * whether mi_tiff_read_box can decode it (*fast = 1).  MI_ERR_INVALID when the file cannot be opened or is not a TIFF. */
int mi_tiff_info(const char* path, int* nx, int* ny, int* dtype, int* fast);

/* out[k][y - y0][x - x0] = slice paths[k] at (y, x) for y in [y0, y1), x in [x0, x1): a box of n slices, decoded on n_threads
 * threads (<= 0: all cores), only the strips the rows touch.  Every file must be fast (mi_tiff_info), of extents nx x ny and
 * sample type dtype; out holds n * (y1 - y0) * (x1 - x0) samples.   [load_bl_tif.cpp: load_bl_tif(files, y, x, height, width)] */
int mi_tiff_read_box(const char* const* paths, int n, int nx, int ny, int dtype, int y0, int y1, int x0, int x1, void* out,
                     int n_threads);

/* One file per z slice of vol [nz][ny][nx]: paths[k] <- slice k, written under a temporary name and renamed when complete; a path
 * that already exists is left alone (LsDeconv.m:1120-1132) -- *written (may be NULL) counts the files produced.
 * compression: 0 none, 1 Adobe deflate at `level` (1 .. 9; save_bl_tif.cpp uses 1).  n_threads <= 0: all cores.
 * [save_bl_tif.cpp: save_bl_tif(volume, fileList, isXYZ, compression, nThreads, useTiles = false)] */
int mi_tiff_write_series(const char* const* paths, int nz, const void* vol, int dtype, int nx, int ny, int compression, int level,
                         int n_threads, int* written);

/* The same files from a volume that lies in DEVICE memory (vol [nz][ny][nx] on device `dev`), deflated there: every strip is one
 * dynamic-Huffman block without string matching (csrc/tiffio.hip: histogram kernel, codes built on the host, encode kernel), the host
 * only frames the streams and writes the files.  Adobe deflate always; readable by any inflater.  Enqueues on `stream` and
 * synchronises it.  No counterpart in the reference (save_bl_tif.cpp compresses on the host's cores). */
int mi_tiff_write_series_device(int dev, void* stream, const char* const* paths, int nz, const void* vol, int dtype, int nx, int ny,
                                int n_threads, int* written);

/* Name of the deflate implementation in use: "libdeflate" when libdeflate.so.0 could be loaded, else "zlib". */
const char* mi_tiff_codec(void);

#ifdef __cplusplus
}
#endif
#endif
