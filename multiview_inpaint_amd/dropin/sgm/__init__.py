"""Drop-in `sgm` package: the dotted class paths the reference's YAML configs name
(svd_inpaint1/configs/test/svd_f_est_ctrl_simp1.yaml:14-177, scripts/sampling/configs/svd.yaml) resolve
to the MI355X implementation in multiview_inpaint_amd.svd. The denoise-loop modules and the first-stage autoencoder's encode / decode (§8f-2) exist here;
the conditioner and the Lightning engines of the reference are out of scope."""
