"""The transform cases shared by the golden generator and the test (pure data)."""
CASES = [("NormalizeScale", {}), ("NormalizeScale", dict(norm_ord=float("inf"), scaling_factor=3.0)), ("NormalizeAxes", {}),
         ("RandomScale", dict(scales=(0.8, 1.25))), ("RandomTranslateGlobal", dict(translate=0.1)),
         ("RandomRotate", dict(degrees=40, axis=2)), ("RandomNormals", dict(translate=0.05)),
         ("SamplePoints", dict(num=64, include_normals=True)), ("NormalizeArea", {})]
