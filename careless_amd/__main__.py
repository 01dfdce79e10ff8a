from careless_amd.careless import main

main()
