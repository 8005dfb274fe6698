"""Drop-in `simple_knn` package for AD-GS on MI355X: `from simple_knn._C import distCUDA2`
(reference: submodules/simple-knn/ext.cpp:15-17, used at scene/gaussian_model.py:20,277)."""
