#!/bin/sh
# ORACLE support (test infrastructure): builds the part of the REFERENCE that compiles from its own sources with gcc alone - the
# host-side, device-independent C files (entity defaults, .lum v4 parser, Wavefront reader, host math, arrays / queues / ring buffers,
# paths) - into oracle/_ref/libluminary_ref_host.so. Nothing is copied: the sources are compiled where they lie under /root/reference.
# The device layer (every file that reaches device_nv_includes.h -> cuda.h / optix.h), mesh.c, image.c (stb), png.c / qoi.c (zlib
# submodule) and luminary.c (generated config.h) are NOT buildable here and are left out; calls into them stay unresolved and are never
# made by the tests. The three -include flags name standard C headers the reference's build gets through its own compile options.
set -e
REF=${LUMINARY_REFERENCE:-/root/reference}
SRC="$REF/src/luminary"
OUT="$(cd "$(dirname "$0")" && pwd)/_ref"
[ -d "$SRC" ] || { echo "reference sources not present: nothing to build"; exit 0; }
mkdir -p "$OUT"
cd "$SRC"
FILES="array.c bvh.c camera.c cloud.c cond_var.c error.c fog.c hashmap.c host_math.c host_memory.c log.c material.c mutex.c name_strings.c
       ocean.c particles.c path.c queue.c queue_worker.c ringbuffer.c sample_count.c scene.c settings.c shared_object.c sky.c texture.c
       thread.c thread_status.c vault_object.c host/host_output_handler.c host/lum.c host/lum_v4.c host/lum_v5.c host/wavefront.c"
gcc -std=gnu11 -O1 -fPIC -shared -w -Wl,-Bsymbolic -include stddef.h -include stdint.h -include stdbool.h -include string.h \
    -I. -Idevice -Ihost -I"$REF/include" -I"$REF/include/luminary" $FILES -o "$OUT/libluminary_ref_host.so" -lm -lpthread
echo "built $OUT/libluminary_ref_host.so"
