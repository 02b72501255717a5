"""Drop-in mirror of Foreground_Instance_Colorization/obj_lib (same module, function and argument
names) over the HIP path.  The reference builds a symbolic TF1 graph and runs it in a session; here the
same calls execute eagerly on the MI355X and return device tensors."""
