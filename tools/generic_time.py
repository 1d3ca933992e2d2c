"""kernel time of the generic render_kernel instantiation (strategy / integrator read at run time) for a few configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ky_amd import api, _abi as A
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
veach = api.mis_scene(1280, 720)
out = []
for name, sc, w, h, kw in (("cornell light_mis", scene, 1024, 768, dict(direct_sample=A.DIRECT_LIGHT_MIS)),
                           ("cornell bsdf", scene, 1024, 768, dict(direct_sample=A.DIRECT_BSDF)),
                           ("cornell recursion_defered", scene, 1024, 768, dict(integrator=A.INTEGRATOR_PATH_TRACING_RECURSION_DEFERED)),
                           ("cornell direct_lighting", scene, 1024, 768, dict(integrator=A.INTEGRATOR_DIRECT_LIGHTING)),
                           ("cornell debug sampler", scene, 1024, 768, dict(sampler=A.SAMPLER_DEBUG)),
                           ("veach light_mis", veach, 1280, 720, dict(direct_sample=A.DIRECT_LIGHT_MIS)),
                           ("veach bsdf_mis", veach, 1280, 720, dict(direct_sample=A.DIRECT_BSDF_MIS)),
                           ("veach recursion", veach, 1280, 720, dict(integrator=A.INTEGRATOR_PATH_TRACING_RECURSION)),
                           ("cornell environment light_mis", api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_ENVIRONMENT, 1024, 768), 1024, 768, dict(direct_sample=A.DIRECT_LIGHT_MIS))):
    p = api.make_params(w, h, 64, **kw)
    api.render(sc, p); api.render(sc, p)
    out.append("%s %.2f" % (name, api.kernel_ms()))
print(" | ".join(out))
