"""TEST INFRASTRUCTURE ONLY — compile the reference's two Cython data helpers from the sources where they lie
(/root/reference/fairseq/data/{data_utils_fast,token_block_utils_fast}.pyx; the reference ships them unbuilt) into
oracle/_ref/cy_lib, so that oracle/gen_golden.py can run fairseq's own batch planner and token-block slicing when it
generates fixtures.  Nothing is written outside oracle/_ref/ (git-ignored); only the build container has /root/reference.

    python oracle/build_ref_cython.py     ->  oracle/_ref/cy_lib/fairseq/data/*.so
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")


def build():
    import numpy as np
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution

    names = ["data_utils_fast", "token_block_utils_fast"]
    if all(any(f.startswith(n) and f.endswith(".so") for f in os.listdir(os.path.join(OUT, "cy_lib", "fairseq", "data")))
           for n in names) if os.path.isdir(os.path.join(OUT, "cy_lib", "fairseq", "data")) else False:
        return os.path.join(OUT, "cy_lib")
    exts = [Extension(f"fairseq.data.{n}", [os.path.join(REF, "fairseq", "data", n + ".pyx")], language="c++",
                      include_dirs=[np.get_include()]) for n in names]
    exts = cythonize(exts, build_dir=os.path.join(OUT, "cy_build"), language_level=3, quiet=True)
    dist = Distribution({"ext_modules": exts})
    cmd = dist.get_command_obj("build_ext")
    cmd.build_lib = os.path.join(OUT, "cy_lib")
    cmd.build_temp = os.path.join(OUT, "cy_build")
    cmd.ensure_finalized()
    cmd.run()
    return cmd.build_lib


def attach():
    """Make the built helpers importable as fairseq.data.* next to the reference's pure-Python package."""
    lib = build()
    import fairseq.data as fd  # the reference package (sys.path is set up by the caller)

    fd.__path__.append(os.path.join(lib, "fairseq", "data"))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("no /root/reference here: nothing to build")
    print(build())
