import os, numpy as np
from PIL import Image
root="gpurun_out/fake_potsdam"
rng=np.random.RandomState(0)
for sub in ("train","test"):
    os.makedirs(os.path.join(root,sub),exist_ok=True); os.makedirs(os.path.join(root,sub+"_convert_labels"),exist_ok=True)
    for i in range(24 if sub=="train" else 4):
        Image.fromarray(rng.randint(0,256,(256,256,3),dtype=np.uint8)).save(os.path.join(root,sub,"%d.tif"%i))
        Image.fromarray(rng.randint(0,6,(256,256),dtype=np.uint8)).save(os.path.join(root,sub+"_convert_labels","%d.png"%i))
