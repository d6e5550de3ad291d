"""Drop-in replacement of the `simple_knn` package (knn/): `from simple_knn._C import distCUDA2`."""
