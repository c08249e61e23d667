#!/usr/bin/env python3
"""Writes a bench workload's triangles as the device scene holds them (3 x float4 per triangle) for tools/bvh_quality.cpp:
  python tools/dump_mesh.py hall /tmp/hall_verts.f32        (no GPU needed: the host layer encodes the scene)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name, path = sys.argv[1], sys.argv[2]
host = bench.build_workload(name, 1920, 1080, 8)
view = host.device_scene()
offsets = np.ctypeslib.as_array(C.cast(view.mesh_tri_offset, C.POINTER(C.c_uint32)), (view.num_meshes + 1,))
largest = int(np.argmax(np.diff(offsets)))  # the big mesh of the workload (instances of small meshes are not what the builders are judged on)
t0, t1 = int(offsets[largest]), int(offsets[largest + 1])
verts = np.ctypeslib.as_array(C.cast(view.vertices, C.POINTER(C.c_float)), (int(offsets[-1]) * 12,))
verts[t0 * 12:t1 * 12].astype(np.float32).tofile(path)
print("%s: mesh %d, %d triangles -> %s" % (bench.WORKLOADS[name], largest, t1 - t0, path))
