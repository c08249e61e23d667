/* Embeds the blue-noise mask into the shared library (the reference embeds its data files with its `Ceb` tool,
 * src/luminary/CMakeLists.txt:206-226). LUM_BLUENOISE_PATH is set by luminary_amd/build.py. */
    .section .rodata
    .balign 16
    .global lum_embedded_bluenoise_2d
    .global lum_embedded_bluenoise_2d_end
lum_embedded_bluenoise_2d:
    .incbin LUM_BLUENOISE_PATH
lum_embedded_bluenoise_2d_end:
    .section .note.GNU-stack,"",@progbits
