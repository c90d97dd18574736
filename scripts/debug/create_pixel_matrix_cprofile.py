"""cProfile of create_pixel_matrix's CALLING thread (the one that drives the GPU): where its ~57 ms per FOV go."""
import cProfile, os, pstats, shutil, sys, tempfile, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ark_analysis_amd import image_io
from ark_analysis_amd.phenotyping import pixie_preprocessing as pp
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 20
root = tempfile.mkdtemp(prefix="pxsom_pre_")
tiff_dir, seg_dir = os.path.join(root, "tiffs"), os.path.join(root, "seg")
os.makedirs(os.path.join(root, "pixel_output_dir")); os.mkdir(seg_dir)
fovs = ["fov%d" % i for i in range(nf)]; chans = ["chan%d" % i for i in range(22)]
rs = np.random.RandomState(0)
for fov in fovs:
    os.makedirs(os.path.join(tiff_dir, fov, "TIFs"))
    for ch in chans:
        img = rs.gamma(0.5, 2.0, size=(1024, 1024)).astype(np.float32); img[rs.uniform(size=img.shape) < 0.4] = 0
        image_io.write_channel(os.path.join(tiff_dir, fov, "TIFs", ch + ".tiff"), img)
    image_io.write_channel(os.path.join(seg_dir, fov + "_whole_cell.tiff"), rs.randint(0, 2000, size=(1024, 1024)).astype(np.int32))
# warm the device / library on a base directory of its own
warm = tempfile.mkdtemp(prefix="pxsom_warm_")
os.makedirs(os.path.join(warm, "pixel_output_dir"))
pp.create_pixel_matrix(fovs[:2], chans, warm, tiff_dir, seg_dir)
shutil.rmtree(warm)
pr = cProfile.Profile()
pr.enable()
pp.create_pixel_matrix(fovs, chans, root, tiff_dir, seg_dir)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
shutil.rmtree(root)
