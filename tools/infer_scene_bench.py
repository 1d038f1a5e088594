import sys, time, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from types import SimpleNamespace
from hypelcnn_amd.classify import train_for_classification as T, infer_for_classification as I
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
alg = os.path.join(root, "hypelcnn_amd/nnmodel/modelconfigs/alg_param_hypelcnn.json")
D = "/tmp/infer_bench"; os.system(f"rm -rf {D}")
scene = "grss2013:h=349:w=1905"   # the real GRSS2013 scene size: 664 845 pixels
argv = ["--loader_name", "SyntheticDataLoader", "--path", scene, "--neighborhood", "3", "--model_name", "HYPELCNNModel",
        "--algorithm_param_path", alg, "--batch_size", "1024", "--step", "20", "--base_log_path", D + "/log",
        "--perform_validation", "false", "--save_checkpoint_steps", "1000", "--importer_name", "GeneratorImporter"]
flags, _ = T.build_parser().parse_known_args(argv)
log_dir = os.path.join(flags.base_log_path, T.get_log_suffix(flags))
T.perform_an_episode(flags, json.load(open(alg)) | {"batch_size": 1024}, T.get_model_from_name("HYPELCNNModel"), log_dir)
for bs in (1024, 4096):
    t0 = time.time()
    r = I.main(["--loader_name", "SyntheticDataLoader", "--path", scene, "--neighborhood", "3", "--model_name",
                "HYPELCNNModel", "--algorithm_param_path", alg, "--batch_size", str(bs), "--base_log_path", log_dir,
                "--output_path", D + "/out", "--domain", "all"])
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"full-scene inference batch {bs}: {r.size} pixels in {dt:.2f} s wall (incl. scene synthesis, graph capture, TIFF) "
          f"= {r.size / dt / 1e3:.1f} k pixels/s")
