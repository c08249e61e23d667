/* Embeds the blue-noise masks (2D: sampler, 1D: output dither), the moon textures and the bridge sampler's table into the shared library (the reference embeds its data files with its `Ceb` tool,
 * src/luminary/CMakeLists.txt:206-226). LUM_BLUENOISE_PATH / LUM_BLUENOISE_1D_PATH are set by luminary_amd/build.py. */
    .section .rodata
    .balign 16
    .global lum_embedded_bluenoise_2d
    .global lum_embedded_bluenoise_2d_end
lum_embedded_bluenoise_2d:
    .incbin LUM_BLUENOISE_PATH
lum_embedded_bluenoise_2d_end:
    .balign 16
    .global lum_embedded_bluenoise_1d
    .global lum_embedded_bluenoise_1d_end
lum_embedded_bluenoise_1d:
    .incbin LUM_BLUENOISE_1D_PATH
lum_embedded_bluenoise_1d_end:
    .balign 16
    .global lum_embedded_moon_albedo
    .global lum_embedded_moon_albedo_end
lum_embedded_moon_albedo:
    .incbin LUM_MOON_ALBEDO_PATH
lum_embedded_moon_albedo_end:
    .balign 16
    .global lum_embedded_moon_normal
    .global lum_embedded_moon_normal_end
lum_embedded_moon_normal:
    .incbin LUM_MOON_NORMAL_PATH
lum_embedded_moon_normal_end:
    .balign 16
    .global lum_embedded_bridge_lut
    .global lum_embedded_bridge_lut_end
lum_embedded_bridge_lut:
    .incbin LUM_BRIDGE_LUT_PATH
lum_embedded_bridge_lut_end:
    .section .note.GNU-stack,"",@progbits
