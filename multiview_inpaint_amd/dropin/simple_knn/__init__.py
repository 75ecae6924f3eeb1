"""Drop-in for the `simple_knn` plug-in the reference imports (`from simple_knn._C import distCUDA2`,
gs-simp/scene/gaussian_model.py:20). See multiview_inpaint_amd/train_ops.py:distCUDA2."""
