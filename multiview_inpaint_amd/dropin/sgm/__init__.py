"""Drop-in `sgm` package: the dotted class paths the reference's YAML configs name
(svd_inpaint1/configs/test/svd_f_est_ctrl_simp1.yaml:14-177, scripts/sampling/configs/svd.yaml) resolve
to the MI355X implementation in multiview_inpaint_amd.svd. Only the denoise-loop modules exist here
(SURVEY.md §8b); the VAE / conditioner / Lightning engines of the reference are out of scope."""
