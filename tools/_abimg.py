import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
out = []
s = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 256, 192)
out.append(hashlib.sha1(api.render(s, api.make_params(256, 192, 100)).tobytes()).hexdigest()[:12])
m = api.mis_scene(256, 144)
out.append(hashlib.sha1(api.render(m, api.make_params(256, 144, 100)).tobytes()).hexdigest()[:12])
out.append(hashlib.sha1(api.render(m, api.make_params(256, 144, 64, direct_sample=A.DIRECT_LIGHT_MIS)).tobytes()).hexdigest()[:12])
two = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA | A.CB_LIGHT_POINT | A.CB_LIGHT_ENVIRONMENT, 256, 192)
out.append(hashlib.sha1(api.render(two, api.make_params(256, 192, 64)).tobytes()).hexdigest()[:12])
out.append(hashlib.sha1(api.render(two, api.make_params(256, 192, 64, integrator=9)).tobytes()).hexdigest()[:12])
print(os.environ.get("KYHIP_LIB"), out)
