import sys; sys.path.insert(0,'path-tracer_amd')
import pyptgpu as pt
for n in ("cornell","materials","obj-model","large-object"):
    sc = pt.LoadedScene(n, 64, 48); g = pt.Gpu(0,0); g.set_scene(sc); print(n, g.scene_info()); g.close()
