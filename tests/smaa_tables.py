"""The SMAA look-up tables live beside the synthetic textures (raytracing_opengl_amd/smaa_tables.py) so that bench.py and the tools can
use them too; tests keep importing this name."""
from raytracing_opengl_amd.smaa_tables import area_table, library_search_table, search_table, synthetic_area_table  # noqa: F401
