"""Triangle clusters and their face planes (PTGPU_OPT_RAY_BINS) as host-side data: ptgpu_cluster_info needs no GPU."""
import numpy as np

import pyptgpu as pt
from test_gpu_synthetic import convex_scene, flat_hexagon, make_scene, room, triangles, _box


def test_only_clusters_with_a_volume_are_tested_by_their_face_planes():
    # cube 6 planes, octahedron 8, prism 5; the open box, the dented cube and the flat double-sided hexagon keep
    # their triangles (0).  The hexagon passes the edge count (every edge used twice) and the vertex test (every
    # vertex in every face plane): what rejects it is that its centroid is not strictly inside its planes.
    sc = make_scene(convex_scene(), 8, 8)
    n, planes = pt.cluster_info(sc)
    assert n == 6
    assert sorted(planes[:n]) == [0, 0, 0, 5, 6, 8], planes
    # the hexagon alone, in every orientation the rounding of its plane offsets could favour
    rng = np.random.default_rng(3)
    for _ in range(50):
        hexagon = flat_hexagon(center=rng.uniform((-30, 10, 40), (30, 70, 120)), radius=rng.uniform(2, 15))
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        c = hexagon.reshape(-1, 3).mean(0)
        hexagon = ((hexagon - c) @ q.T.astype(np.float32) + c).astype(np.float32)
        n, planes = pt.cluster_info(make_scene(room() + [triangles(hexagon)], 8, 8))
        assert n == 1 and planes[0] == 0, planes
    # a real box still qualifies
    n, planes = pt.cluster_info(make_scene(room() + [triangles(_box((-10, 0, 60), (10, 22, 80)))], 8, 8))
    assert n == 1 and planes[0] == 6


def test_cornell_cube_is_a_convex_cluster():
    sc = pt.LoadedScene("cornell", 8, 8)
    n, planes = pt.cluster_info(sc)
    assert n >= 1 and 6 in planes[:n]
